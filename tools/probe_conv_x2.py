"""fp16x2 3 x 3 convolution (vd3d_conv3x3_x2) on the DPT neck / head shapes of DA-V2-Base at 4K (16 frames): time beside the float32 library convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd.render_3d import Renderer
F = torch.nn.functional


def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


R = Renderer(0)
torch.backends.cudnn.benchmark = True
g = torch.Generator(device="cuda").manual_seed(1)
for (H, W, Cin, Cout) in ((37, 66, 128, 128), (74, 132, 128, 128), (148, 264, 128, 128), (296, 528, 128, 64), (19, 33, 768, 128), (148, 264, 96, 128), (518, 924, 64, 32)):
    B = 16
    x = torch.relu(torch.randn(B, Cin, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
    img = R.conv3x3_x2_pack(w)
    y = R.conv3x3_x2(x, img, Cout)
    y32 = F.conv2d(x, w, None, 1, 1)
    ref = F.conv2d(x[:1].double(), w.double(), None, 1, 1)
    r3 = float((y[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    r32 = float((y32[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    t3 = bench(lambda: R.conv3x3_x2(x, img, Cout))
    t32 = bench(lambda: F.conv2d(x, w, None, 1, 1))
    fl = 2.0 * B * H * W * Cin * 9 * Cout
    print(f"{H}x{W} {Cin}->{Cout}: x2 {t3:.3f} ms = {fl / t3 / 1e9:.0f} TF-equiv | library f32 {t32:.3f} ms = {fl / t32 / 1e9:.0f} TF | rel rms x2 {r3:.2e} f32 {r32:.2e}", flush=True)
