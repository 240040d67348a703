"""MI355X-native batched balance-controller solve."""
