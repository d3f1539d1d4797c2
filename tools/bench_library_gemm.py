#!/usr/bin/env python3
"""Reference point for the encoder's GEMM shapes: what the vendor library (hipBLASLt / rocBLAS through
torch.matmul, fp16 in, fp32 accumulate, NO epilogue fused) takes for the same M x N x K products at the 64 k-token
batch.  Not used by the product; DESIGN.md quotes the numbers beside the hand-written kernels'."""
import json
import torch

M = 65536
SHAPES = {"QKV": (1152, 384), "out-projection": (384, 384), "FFN1": (1536, 384), "FFN2": (384, 1536)}


def main():
    dev = torch.device("cuda:0")
    out = {}
    for name, (N, K) in SHAPES.items():
        x = torch.randn((M, K), device=dev, dtype=torch.float16)
        w = torch.randn((N, K), device=dev, dtype=torch.float16) * 0.03
        b = torch.randn((N,), device=dev, dtype=torch.float16)
        for _ in range(5):
            y = torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            y = torch.nn.functional.linear(x, w, b)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 50 * 1e3
        out[name] = {"M": M, "N": N, "K": K, "us": round(us, 1), "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1)}
        if name == "FFN1":
            for _ in range(3):
                z = torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))
            torch.cuda.synchronize()
            t0.record()
            for _ in range(50):
                z = torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))
            t1.record()
            torch.cuda.synchronize()
            out["FFN1 + separate GELU kernel"] = {"us": round(t0.elapsed_time(t1) / 50 * 1e3, 1)}
        del x, w, b, y
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
