"""Static check for inline-asm LDS reads whose wait sits in a LATER asm statement (kz_fpaq.hip: FPW_DEC_OPEN issues ds_read_b32
into c0 / c1, the s_waitcnt is in the next FPW_DEC_BIT2).  The compiler does not track LDS operations issued from inline asm, so
nothing stops it from placing a copy or spill of those registers between the two statements; this walks the device assembly of a
kernel in layout order with the queue of outstanding LGKM operations (LDS returns in order; `s_waitcnt lgkmcnt(N)` leaves the
youngest N outstanding) and reports every instruction that touches the destination of a ds_read that has not been waited for.
   python tools/asm_lds_hazards.py [kernel ...]      (compiles kanzi_amd/csrc/kz_fpaq.hip to assembly: about a second)
Used by tests/test_abi.py on every CPU test run, i.e. with whatever hipcc builds the library."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_KERNELS = ("k_fpaq_dec_wave2", "k_fpaq_enc_wave")


def device_asm(source, hipcc=None):
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", "-o", out, source],
                       check=True, capture_output=True)
        return open(out).read()


def kernel_body(text, name):
    m = re.search(r"^(_Z\d+%s[A-Z]\w*):" % re.escape(name), text, re.M)
    if not m:
        raise KeyError(name)
    return text[m.end():text.index(".Lfunc_end", m.end())]


def _vgprs(operands):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", operands):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def hazards(body):
    """-> (violations [(line, instruction, registers)], number of ds reads seen)"""
    queue, bad, reads = [], [], 0
    for ln, line in enumerate(body.splitlines()):
        s = line.split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            continue
        op, _, args = s.partition(" ")
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", args)
            if m:
                n = int(m.group(1))
                if any(kind == "smem" for kind, _ in queue):       # scalar loads may return out of order: only 0 says anything
                    if n == 0:
                        queue = []
                else:
                    del queue[:max(0, len(queue) - n)]
            continue
        used = _vgprs(args)
        for _, dest in queue:
            if dest & used:
                bad.append((ln, s, sorted(dest & used)))
        if op.startswith("ds_read") or op.startswith("ds_load"):
            queue.append(("lds", _vgprs(args.split(",")[0])))
            reads += 1
        elif op.startswith("ds_"):
            queue.append(("lds", set()))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append(("smem", set()))
    return bad, reads


if __name__ == "__main__":
    text = device_asm(os.path.join(ROOT, "kanzi_amd", "csrc", "kz_fpaq.hip"))
    rc = 0
    for name in (sys.argv[1:] or DEFAULT_KERNELS):
        bad, reads = hazards(kernel_body(text, name))
        print("%-20s %2d ds reads, %d unwaited uses" % (name, reads, len(bad)))
        for b in bad[:10]:
            print("   line %d: %s  (v%s)" % (b[0], b[1], ",v".join(map(str, b[2]))))
        rc |= bool(bad)
    sys.exit(rc)
