# does a whole pipeline step replayed as ONE HIP graph (captured from the context's stream) beat the 35 separate launches?
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereo_visual_slam_amd.pipeline import KeyframePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pipe = KeyframePipeline(B, anms_num=1500, unique_frames=min(B, 64), ba_windows="tracks")
for _ in range(3): pipe.step()
pipe.vo.sync(); torch.cuda.synchronize()
def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    pipe.vo.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
t_plain = timed(pipe.step)
ref = pipe.ba_T.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=pipe.stream):
        pipe.step()
    g.replay(); torch.cuda.synchronize()
    t_graph = timed(g.replay)
    same = bool(torch.equal(ref, pipe.ba_T))
    print("B=%d  plain %.3f ms/step  graph %.3f ms/step  results identical: %s" % (B, t_plain, t_graph, same))
except Exception as e:
    print("capture failed:", repr(e)[:500])
pipe.close()
