"""Wave executor A/B on one box: one-lane handle vs ESAHRNET_STREAMS=4, graph replay and eager, alternating rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esa_pose_estimation_amd import config, seg_hrnet2, synth, inference

def make():
    net = seg_hrnet2.get_seg_model(config.make_config())
    net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
    return net.cuda().eval()

def timed(fn, n=100):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

x = torch.randn(32, 1, 256, 256, device="cuda")
nets, graphs, outs = {}, {}, {}
with torch.no_grad():
    for lanes in ("1", "4"):
        os.environ["ESAHRNET_STREAMS"] = lanes
        net = nets[lanes] = make()
        ref = net(x).clone(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): inference.heatmaps_to_keypoints(net(x))
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = graphs[lanes] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            h = net(x); kp = inference.heatmaps_to_keypoints(h)
        g.replay(); torch.cuda.synchronize()
        outs[lanes] = (h.clone(), kp.clone())
        print(lanes, "graph == eager:", torch.equal(h, ref), flush=True)
    print("lanes 4 == lanes 1:", torch.equal(outs["1"][0], outs["4"][0]), torch.equal(outs["1"][1], outs["4"][1]))
    for r in range(4):
        for lanes in ("1", "4"):
            print(f"round {r} lanes {lanes}: graph {timed(graphs[lanes].replay):.4f} ms   eager {timed(lambda: nets[lanes](x), 40):.4f} ms", flush=True)
