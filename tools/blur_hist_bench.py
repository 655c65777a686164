"""to_uint8 + blur + histogram of 64 planes of 4096^2 uint16 on their own: the one-pass kernel (mg_to_uint8_blur_hist)
against the two passes behind the same entry point (MG_NO_BLUR_HIST=1 in another process).  HIP events, best of 7.
python tools/blur_hist_bench.py [--planes 64] [--size 4096]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from magnify_amd import _native as nat, hotpath as hp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    P, S = args.planes, args.size
    g = torch.Generator(device="cuda").manual_seed(1)
    d = (torch.rand((P, S, S), device="cuda", generator=g) * 3000 + 200).to(torch.uint16)
    d[:, ::64, ::64] = 40000
    mm = hp.plane_minmax(d)
    lib, s = nat.lib(), torch.cuda.current_stream().cuda_stream
    words = int(lib.mg_blur_hist_scratch_words(P, S, S))
    scratch = torch.empty((words,), dtype=torch.int32, device="cuda")
    blur = torch.zeros((P, S, S), dtype=torch.uint8, device="cuda")
    hist = torch.zeros((P, 12288), dtype=torch.int32, device="cuda")

    def run():
        nat.check(lib.mg_to_uint8_blur_hist(d.data_ptr(), nat.dtype_code(d.dtype), P, d.stride(0), S, S, d.stride(1), mm.data_ptr(),
                                            blur.data_ptr(), 0, hist.data_ptr(), scratch.data_ptr(), words, s), "blur_hist")
    ts = []
    for _ in range(args.reps):
        hist.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    print(json.dumps({"one_pass": os.environ.get("MG_NO_BLUR_HIST") is None, "ms_best": round(min(ts), 4),
                      "ms_median": round(sorted(ts)[len(ts) // 2], 4), "hist_sum_ok": bool((hist.sum(dim=1) == S * S).all()),
                      "checksum": int(blur.view(torch.int8).to(torch.int64).sum().item()) ^ int(hist.to(torch.int64).mul(torch.arange(12288, device="cuda")).sum().item())}))


if __name__ == "__main__":
    main()
