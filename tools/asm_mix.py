"""Instruction mix of one kernel from a device-only assembly listing (hipcc -S --cuda-device-only):
    python tools/asm_mix.py <file.s> <substring of the mangled kernel name> [--loop]
Counts by mnemonic, with the quarter-rate VALU classes (32-bit integer multiplies, transcendental / float64 rcp,
sqrt) summed separately; --loop restricts the count to the largest backward-branch loop body."""
import collections
import re
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    text = open(path).read()
    m = None
    for m_ in re.finditer(r"^(\S+):\s*;\s*@\1\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        if needle in m_.group(1):
            m = m_
            break
    if m is None:
        raise SystemExit(f"no kernel matching {needle}")
    lines = [ln.split(";")[0].strip() for ln in m.group(2).splitlines()]
    if "--loop" in sys.argv:
        labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
        best = None
        for i, ln in enumerate(lines):
            mm = re.match(r"s_(?:cbranch_\w+|branch)\s+(\S+)", ln)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                span = (labels[mm.group(1)], i)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        lines = lines[best[0]: best[1] + 1]
    ins = [ln.split()[0] for ln in lines if ln and not ln.startswith((";", ".")) and not ln.endswith(":")]
    c = collections.Counter(ins)
    quarter = sum(v for k, v in c.items() if re.match(r"v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_lo_i32|mad_u64_u32|mad_i64_i32|rcp_f64|rsq_f64|sqrt_f64|rcp_f32|rsq_f32|sqrt_f32|exp_f32|log_f32|sin_f32|cos_f32)", k))
    f64 = sum(v for k, v in c.items() if "_f64" in k)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(m.group(1))
    print(f"instructions {len(ins)}  VALU {valu}  (float64 {f64}, quarter-rate {quarter})  SALU {sum(v for k, v in c.items() if k.startswith('s_'))}"
          f"  LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}  global/flat/buffer {sum(v for k, v in c.items() if k.startswith(('global_', 'flat_', 'buffer_')))}")
    for k, v in c.most_common(40):
        print(f"  {k:28s} {v}")


if __name__ == "__main__":
    main()
