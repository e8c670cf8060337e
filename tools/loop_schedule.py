"""Print a GEMM main loop as "what sits behind MFMA slot n" — used to compare the schedule of gemm_w4_kernel with the vendor library's
MT256x256x64 kernel (r06; profiles/r06_gemm_schedules.txt).  Input: hipcc -S output (our kernels) or llvm-objdump -d output (the vendor's
code object, unbundled with clang-offload-bundler).  Finds every backward-branch loop holding exactly 128 v_mfma and prints the first.

  python tools/loop_schedule.py asm   gemm_dev.s _ZN12_GLOBAL__N_114gemm_w4_kernelILi0ELb1ELi0EEEvNS_8GemmArgsE
  python tools/loop_schedule.py objdump custom.s
"""
import re
import sys


def classify(op, args):
    if op.startswith("ds_read"):
        off = args.split("offset:")[1] if "offset:" in args else "0"
        return f"R {args.split(',')[0].strip()} +{off}"
    if op.startswith("buffer_load"):
        return "DMA " + " ".join(args.split(",")[:1]).strip()
    if op == "s_waitcnt":
        return "WAIT " + args
    if op == "s_barrier":
        return "BARRIER"
    return (op + " " + args)[:44]


def emit(stream):
    k, row = 0, []
    for op, args in stream:
        if op.startswith("v_mfma"):
            if row:
                print(f"  after mfma {k - 1:3d}: " + " ; ".join(row))
                row = []
            k += 1
        else:
            row.append(classify(op, args))
    if row:
        print(f"  after mfma {k - 1:3d}: " + " ; ".join(row))


def from_asm(path, symbol):
    text = open(path).read()
    a = text.index(symbol + ":")
    body = text[a:text.index(".end_amdhsa_kernel", a)].split("\n")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            seg = body[labels[m.group(1)]:i + 1]
            if sum("v_mfma" in x for x in seg) == 128:
                stream = []
                for x in seg:
                    x = x.split(";")[0].strip()
                    if x and not x.startswith("."):
                        parts = x.split(None, 1)
                        stream.append((parts[0], parts[1] if len(parts) > 1 else ""))
                return emit(stream)
    raise SystemExit("no 128-MFMA loop found")


def from_objdump(path):
    ins = []
    for l in open(path):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((m.group(1), m.group(2)))
    start, cnt = 0, 0
    for i, (op, args) in enumerate(ins):
        cnt += op.startswith("v_mfma")
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            if cnt == 128:
                return emit(ins[start:i + 1])
            start, cnt = i + 1, 0
    raise SystemExit("no 128-MFMA loop found")


if __name__ == "__main__":
    from_asm(sys.argv[2], sys.argv[3]) if sys.argv[1] == "asm" else from_objdump(sys.argv[2])
