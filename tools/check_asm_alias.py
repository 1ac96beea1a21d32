"""Static check of the generated code for the hand-issued LDS reads of the matcher's score kernel:
no `ds_read_*` may name, as its address, a register inside the destination of a read issued before
it in the same inline-asm block (the first read's data would overwrite the address while the
second read may still be waiting to issue -- a rare wrong operand, not a fault).

  python3 tools/check_asm_alias.py            compiles visual_slam_amd/csrc/match.hip to assembly and checks it
  python3 tools/check_asm_alias.py file.s     checks an assembly file
Exit status 1 when a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ = re.compile(r"\s*ds_read\w*\s+v\[(\d+):(\d+)\],\s*v(\d+)")


def hazards(text):
    """(line number, first read, second read) for every address taken from an in-flight destination."""
    out = []
    block = []          # (lo, hi, line text) of the reads of the current inline-asm block
    inside = False
    for n, line in enumerate(text.split("\n"), 1):
        if "#ASMSTART" in line:
            inside, block = True, []
        elif "#ASMEND" in line:
            inside = False
        elif inside:
            m = READ.match(line)
            if m:
                lo, hi, addr = map(int, m.groups())
                for plo, phi, ptxt in block:
                    if plo <= addr <= phi:
                        out.append((n, ptxt.strip(), line.strip()))
                block.append((lo, hi, line))
    return out


def compile_to_asm(src):
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "out.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "visual_slam_amd", "csrc"),
               "-Wno-unused-function", src, "-o", s]
        subprocess.run(cmd, check=True, capture_output=True)
        return open(s).read()


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else compile_to_asm(os.path.join(ROOT, "visual_slam_amd", "csrc", "match.hip"))
    n_reads = len([l for l in text.split("\n") if READ.match(l)])
    bad = hazards(text)
    for n, a, b in bad:
        print("line %d: `%s` takes its address from the destination of `%s`" % (n, b, a))
    print("%d wide LDS reads checked, %d hazards" % (n_reads, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
