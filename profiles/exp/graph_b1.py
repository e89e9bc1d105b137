"""Does replaying the B = 1 forward as one captured hipGraph beat launching its kernels one by one?
python3 profiles/exp/graph_b1.py [HINT|AST] [size] [batch]"""
import os
import sys
import time

import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "image-restoration_amd"))
from jirf_hip.synthetic import default_config, make_state, synthetic_images  # noqa: E402
from basicsr.models.archs import define_network  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "HINT"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = default_config(arch)
net = define_network(dict(cfg, type=arch))
net.load_state_dict(make_state(arch, cfg, seed=1))
net.cuda(0).eval()
net.batch_mode = "per_image"
x = torch.from_numpy(synthetic_images(B, size, size, seed=7, sigma=50 if arch == "HINT" else 0)).cuda()


def step():
    if arch == "HINT":
        net.qv_cache = None
    return net(x)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


y_ref = step().clone()
print(f"{arch} B={B} {size}x{size}: eager {timed(step):.3f} ms")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
    with torch.cuda.graph(g, stream=side):
        y = step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(f"graph replay {timed(g.replay):.3f} ms, max |diff| vs eager {float((y - y_ref).abs().max()):.3e}")
