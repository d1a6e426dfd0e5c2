"""Per-kernel derived figures from a PMC summary JSON (tools/pmc_summary.py --json): duration, HBM traffic as a share of
8 TB/s, VALU issue share 4 * SQ_INSTS_VALU / (1024 SIMDs * cycles), LDS-array share SQ_LDS_IDX_ACTIVE / (256 CUs * cycles),
LDS bank-conflict share, mean waves per SIMD (SQ_WAVE_CYCLES is counted in 4-cycle units), and which of them is largest.
cycles = duration of the dispatch in the counter pass x 2.4 GHz (a lower bound on the shares: profiled passes clock lower).

Usage: python tools/kernel_limits.py profiles/r03_pmc.json > profiles/r03_<tag>_kernel_limits.txt"""
import json, sys

d = json.load(open(sys.argv[1]))["kernels"]
rows = []
for k, c in d.items():
    ns = c.get("duration_ns")
    if not ns:
        continue
    cyc = ns * 2.4
    hbm = (c.get("hbm_bytes_per_launch") or 0.0) / (ns * 1e-9) / 8e12
    valu = 4.0 * c.get("SQ_INSTS_VALU", 0.0) / (1024.0 * cyc)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc)
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0))
    waves = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (1024.0 * cyc)
    top = max((hbm, "hbm"), (valu, "valu"), (lds, "lds"))
    rows.append((ns, k, hbm, valu, lds, conf, waves, top[1] if top[0] > 0.5 else "latency / occupancy"))
tot = sum(r[0] for r in rows)
print("%-20s %9s %6s %6s %6s %6s %9s %10s  %s" % ("kernel", "us", "share", "hbm", "valu", "lds", "lds-confl", "waves/SIMD", "largest (> 0.5)"))
for ns, k, hbm, valu, lds, conf, waves, top in sorted(rows, reverse=True):
    print("%-20s %9.1f %6.3f %6.2f %6.2f %6.2f %9.2f %10.1f  %s" % (k, ns / 1e3, ns / tot, hbm, valu, lds, conf, waves, top))
print("sum of kernel durations in the counter pass: %.3f ms" % (tot / 1e6))
