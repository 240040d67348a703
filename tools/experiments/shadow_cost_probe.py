"""What the shadow wavefronts cost the solve they ride with: the placed entry with a fixed order (library's latency placement
of the batch's own counts) with and without prev_iterations / next_robot_order, hipGraph of 20 launches, best of 5 x 10 replays.
usage: shadow_cost_probe.py LIB [LIB ...]"""
import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np


def one(lib):
    from quadruped_locomotion_amd import capi, synth
    capi.LIB_PATH = os.path.abspath(lib)
    import torch
    ctx = capi.Context(device=0)
    out = {}
    for gait, errors, B in (("static", "survey", 4096), ("static", "calm", 4096), ("trot", None, 8192)):
        state = synth.make_states(B, gait, errors=errors)
        d = capi.to_device(state)
        tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
        status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        iters = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        prev = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        nxt = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        ctx.balance_solve_placed_device(d, tau, None, status, iterations=prev, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        order = torch.from_numpy(ctx.placement_from_iterations(prev.cpu().numpy(), policy=capi.PLACEMENT_LATENCY)).to("cuda:0")
        for name, kw in (("order only", {}), ("order + next placement", dict(prev_iterations=prev, next_order=nxt, policy=capi.PLACEMENT_LATENCY))):
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(20):
                        ctx.balance_solve_placed_device(d, tau, None, status, order=order, iterations=iters,
                                                        stream=torch.cuda.current_stream().cuda_stream, **kw)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            best = 1e9
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
            out["%s%s %d %s" % (gait, "-" + errors if errors else "", B, name)] = round(best, 2)
    ctx.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        print(json.dumps(one(sys.argv[2])))
    else:
        for rep in range(3):
            for lib in sys.argv[1:]:
                r = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
                print("%-34s %s" % (lib, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
