#!/usr/bin/env python3
"""Where the wave cycles of the search kernels go, from the committed PMC summary (profiles/collect.sh + summarize.py).

usage: stalls.py profiles/r04_trna [kernel substring ...]   (default: rma_search_kernel rma_drain_kernel)

Units.  The SQ cycle counters of gfx950 tick in quad-cycles (4 shader clocks) per wave or per SIMD: SQ_WAVE_CYCLES /
SQ_WAVES times four is the kernel's length in clocks (checked below against SQ_BUSY_CYCLES / 32 shader engines and the
traced duration), and SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU -- a wave64 VALU instruction holds its SIMD for one
quad-cycle.  A wave's resident time splits into ACTIVE_INST_ANY (an instruction of it is executing), WAIT_INST_ANY (it
has an instruction ready and waits for issue) and WAIT_ANY (it waits for data or a barrier: s_waitcnt, s_barrier); the
three add up to SQ_WAVE_CYCLES within a percent, which is the check that they are read right.

Floors.  Different instruction classes issue side by side from different waves, so the kernel cannot be shorter than
its busiest unit: VALU at one quad-cycle per instruction and SIMD (the rate profiles/valu_peak.hip measures for
independent integer instructions at 4 waves per SIMD is 3.5 clocks, quoted beside it), the scalar unit at one
instruction per clock and SIMD, LDS at two clocks per wave64 access plus the conflict cycles per CU.
"""
import csv
import os
import sys

base = sys.argv[1]
kernels = sys.argv[2:] or ["rma_search_kernel", "rma_drain_kernel"]
SIMDS, CUS, SES = 1024, 256, 32
rows = list(csv.DictReader(open(base + "_pmc_summary.csv")))
dur = {}
ks = base + "_kernel_stats.csv"
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        dur[r["Name"]] = float(r["AverageNs"]) * 1e-6


def counters(sub):
    c = {}
    for r in rows:
        if sub in r["kernel"]:
            c[r["counter"]] = float(r["mean_per_dispatch"])
    return c


for sub in kernels:
    c = counters(sub)
    if not c:
        continue
    ms = next((v for k, v in dur.items() if sub in k), None)
    wc, waves = c["SQ_WAVE_CYCLES"], c["SQ_WAVES"]
    clocks = wc / waves * 4
    ghz = c["SQ_BUSY_CYCLES"] / SES / (ms * 1e-3) / 1e9 if ms else None
    print(f"== {sub}: {waves:.0f} waves, {ms:.3f} ms traced; a wave is resident {clocks / 1e6:.2f} M clocks "
          f"(SQ_BUSY_CYCLES / {SES} = {c['SQ_BUSY_CYCLES'] / SES / 1e6:.2f} M clocks -> {ghz:.2f} GHz)")
    act, wia, wa = c["SQ_ACTIVE_INST_ANY"], c["SQ_WAIT_INST_ANY"], c["SQ_WAIT_ANY"]
    print(f"   of a wave's resident time: executing an instruction {100 * act / wc:5.1f} %   ready, waiting for issue {100 * wia / wc:5.1f} %   "
          f"waiting for data or a barrier {100 * wa / wc:5.1f} %   (sum {100 * (act + wia + wa) / wc:.1f} %)")
    print(f"   ... of which waiting for an LDS instruction to issue (SQ_WAIT_INST_LDS): {100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} %")
    parts = [("VALU", "SQ_ACTIVE_INST_VALU"), ("scalar", "SQ_ACTIVE_INST_SCA"), ("LDS", "SQ_ACTIVE_INST_LDS"), ("branch / barrier / waitcnt (MISC)", "SQ_ACTIVE_INST_MISC"),
             ("flat / global", "SQ_ACTIVE_INST_FLAT")]
    print("   executing, by class (share of resident time): " + ", ".join(f"{n} {100 * c.get(k, 0) / wc:.1f} %" for n, k in parts))
    lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
    print(f"   instructions per launch: VALU {c['SQ_INSTS_VALU'] / 1e6:.1f} M (at {lanes:.1f} of 64 lanes), SALU {c['SQ_INSTS_SALU'] / 1e6:.1f} M, "
          f"LDS {c['SQ_INSTS_LDS'] / 1e6:.1f} M ({c.get('SQ_INSTS_LDS_LOAD', 0) / 1e6:.1f} M loads, {c.get('SQ_INSTS_LDS_STORE', 0) / 1e6:.1f} M stores; "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / 1e6:.1f} M conflict cycles = {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f} per LDS quad-cycle), "
          f"branches {c.get('SQ_INSTS_BRANCH', 0) / 1e6:.1f} M, vector memory {c['SQ_INSTS_VMEM_RD'] / 1e6:.2f} M reads / {c['SQ_INSTS_VMEM_WR'] / 1e6:.2f} M writes, "
          f"instruction fetches {c.get('SQ_IFETCH', 0) / 1e6:.1f} M")
    per_simd = wc / SIMDS * 4            # clocks a SIMD has waves on it, summed over its waves
    occ = wc * 4 / (c["SQ_BUSY_CYCLES"] / SES * SIMDS)
    print(f"   waves resident per SIMD, mean over the kernel: {occ:.2f}")
    if ghz:
        f = ghz * 1e9
        valu4 = c["SQ_INSTS_VALU"] * 4 / SIMDS / f * 1e3
        valu35 = c["SQ_INSTS_VALU"] * 3.5 / SIMDS / f * 1e3
        salu = c["SQ_INSTS_SALU"] / SIMDS / f * 1e3
        lds = (c["SQ_INSTS_LDS"] * 2 + c.get("SQ_LDS_BANK_CONFLICT", 0)) / CUS / f * 1e3
        print(f"   floors of this instruction mix: VALU {valu4:.3f} ms at one quad-cycle per instruction ({valu35:.3f} ms at the measured 3.5 clocks), "
              f"scalar {salu:.3f} ms, LDS {lds:.3f} ms  ->  the kernel runs at {max(valu4, salu, lds) / ms:.2f} of its busiest unit's floor "
              f"({max(valu35, salu, lds) / ms:.2f} by the measured VALU rate)")
