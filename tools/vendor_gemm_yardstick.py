"""Vendor yardstick (tools only, never in the product): time hipBLASLt / rocBLAS through torch.matmul (bf16) on the three contraction
shapes of a C2 layer, with uniform random [-1, 1) operands, so that the tile engine's TFLOP/s have an external reference on the same box.

  gate    : [88000, 848] x [848, 512]     (M = 512 output channels, K = 3*256 + 80)
  d x     : [88000, 1536] x [1536, 256]
  dgate   : [88000, 512] x [512, 256] ... and the weight gradient shape [768, 88000] x [88000, 512] (contraction over time)
"""
import sys, time, torch

def bench(name, a, b, iters=20):
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
    print(f"{name:34s} {tuple(a.shape)} x {tuple(b.shape)}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF", flush=True)

def main():
    dev = torch.device("cuda:0")
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 88000
    g = torch.Generator(device=dev); g.manual_seed(1)
    r = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    print(f"torch {torch.__version__}, {torch.cuda.get_device_name(0)}; bf16, uniform random operands, rows = {rows}")
    for rnd in range(2):
        bench("gate   512 x rows x 848", r(rows, 848), r(848, 512))
        bench("gate^T (W first)", r(512, 848), r(848, rows))
        bench("gate K padded to 896", r(rows, 896), r(896, 512))
        bench("d x    256 x rows x 1536", r(rows, 1536), r(1536, 256))
        bench("d x^T  (W first)", r(256, 1536), r(1536, rows))
        bench("wgrad  768 x 512 x rows", r(768, rows), r(rows, 512))
        bench("square 8192^3", r(8192, 8192), r(8192, 8192), iters=5)

if __name__ == "__main__":
    main()
