"""Static check of the score kernel's generated code (no GPU needed: hipcc cross-compiles).

The kernel issues pairs of LDS reads by hand in one inline-asm statement.  If the compiler gives
the address operand a register inside the first read's destination, the second read takes its
address from a register the first (still in flight) overwrites: harmless most of the time, a wrong
operand when the wave is held up between the two.  That once cost a few wrong nearest neighbours
per 4096 x 4096 pair; the outputs are early-clobber now, and this test keeps it that way."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_checker():
    spec = importlib.util.spec_from_file_location("check_asm_alias", os.path.join(ROOT, "tools", "check_asm_alias.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_sees_the_hazard_pattern():
    chk = load_checker()
    bad = "\t;;#ASMSTART\n\tds_read_b128 v[8:11], v8\n\tds_read_b128 v[12:15], v8 offset:0x2000\n\t;;#ASMEND\n"
    good = "\t;;#ASMSTART\n\tds_read_b128 v[8:11], v203\n\tds_read_b128 v[12:15], v203 offset:0x2000\n\t;;#ASMEND\n"
    assert len(chk.hazards(bad)) == 1 and chk.hazards(good) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_score_kernel_reads_never_take_an_address_from_an_in_flight_destination():
    chk = load_checker()
    text = chk.compile_to_asm(os.path.join(ROOT, "visual_slam_amd", "csrc", "match.hip"))
    assert text.count("ds_read_b128") >= 48          # the kernel's hand-issued fragment reads are there
    assert chk.hazards(text) == []
