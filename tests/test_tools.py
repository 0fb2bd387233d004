"""tools/pmc_report.py on synthetic rocprofv3 counter CSVs (CPU only): passes are joined per dispatch index, a dispatch whose
clock is impossible is rejected instead of averaged (the r03 table's 203 GHz row: ONE of 8 dispatches reported a
GRBM_GUI_ACTIVE 1000x too large), mixed-role kernel names get no traffic ratio, and algorithmic bytes follow the role."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
       "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
       "Counter_Value", "Start_Timestamp", "End_Timestamp"]
P = 32 * 480 * 640
K_GATED = "void codon::conv_c8_kernel<codon::C8Bf16, 5, 64, 64, false, 4, true, false, false>(codon::ConvC8Params)"
K_MIXED = "void codon::conv_c8_kernel<codon::C8Bf16, 3, 64, 64, false, 16, false, true, true>(codon::ConvC8Params)"
K_WG128 = "void codon::conv_wgrad_c8_kernel<codon::C8Bf16, 5, false, 128, 128, false>(codon::WgradC8Params)"


def _write(d, rows):
    os.makedirs(os.path.join(d, "run"), exist_ok=True)
    with open(os.path.join(d, "run", "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(HDR)
        for disp, name, counter, value, ns in rows:
            w.writerow([disp, disp, "Agent 2", 1, 1, 1, 65536, 8, name, 256, 0, 0, 128, 0, 96, counter, value, 1000000 * disp, 1000000 * disp + ns])


def test_pmc_report_joins_per_dispatch_and_rejects_impossible_clocks(tmp_path):
    fe, wr, bu, out = (str(tmp_path / n) for n in ("fetch", "write", "busy", "out.json"))
    seq = [K_GATED, K_MIXED, K_WG128] * 8                        # dispatch ids 1..24, same order in every pass
    ns = {K_GATED: 1_770_000, K_MIXED: 730_000, K_WG128: 5_500_000}
    gb = {K_GATED: 5.62e9, K_MIXED: 2.95e9, K_WG128: 7.6e9}
    _write(fe, [(i + 1, k, "FETCH_SIZE", 0.6 * gb[k] / 2 / 1024, ns[k]) for i, k in enumerate(seq)])     # KB, tallied at half
    _write(wr, [(i + 1, k, "WRITE_SIZE", 0.4 * gb[k] / 1024, ns[k]) for i, k in enumerate(seq)])
    rows = []
    for i, k in enumerate(seq):
        cyc = 1.8 * ns[k]                                        # 1.8 GHz per XCD
        grbm = 8 * cyc * (1000.0 if (k == K_GATED and i == 0) else 1.0)       # the first gated dispatch: 1000x too large
        rows += [(i + 1, k, "GRBM_GUI_ACTIVE", grbm, ns[k]), (i + 1, k, "SQ_VALU_MFMA_BUSY_CYCLES", 0.6 * 1024 * cyc, ns[k])]
    _write(bu, rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_report.py"), fe, wr, bu, out, "32", "480", "640", "train", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ks = json.load(open(out))["kernels"]
    g = ks["conv_c8_kernel<C8Bf16, 5, 64, 64, false, 4, true, false, false>"]
    assert g["launches"] == 8 and g["busy_pass_rejected"] == 1 and "suspect" not in g
    assert abs(g["clock_ghz"] - 1.8) < 1e-6 and abs(g["mfma_busy_frac"] - 0.6) < 1e-6          # the bad dispatch is not averaged in
    assert abs(g["hbm_bytes_per_launch"] - 5.62e9) < 1e3
    assert g["alg_bytes_per_launch"] == (3 * 64 + 64) * 2 * P and "emitted" in g["roles"]       # pre + inputs in, gated + y out
    m = ks["conv_c8_kernel<C8Bf16, 3, 64, 64, false, 16, false, true, true>"]
    assert m["roles"] == "mixed" and "traffic_over_alg" not in m                               # fwd + dgrad variants share the name
    w = ks["conv_wgrad_c8_kernel<C8Bf16, 5, false, 128, 128, false>"]
    assert w["alg_bytes_per_launch"] == 256 * 2 * P and abs(w["traffic_over_alg"] - 7.6e9 / (256 * 2 * P)) < 1e-9
    assert all(v.get("clock_ghz", 1.0) < 2.6 for v in ks.values())


def test_pmc_report_flags_a_pass_mismatch(tmp_path):
    fe, wr, out = (str(tmp_path / n) for n in ("fetch", "write", "out.json"))
    _write(fe, [(i + 1, K_WG128, "FETCH_SIZE", 1e6, 5_000_000) for i in range(4)])
    _write(wr, [(i + 1, K_WG128, "WRITE_SIZE", 1e5, 5_000_000) for i in range(3)])             # one launch fewer
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_report.py"), fe, wr, "-", out, "32", "480", "640", "train", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    w = json.load(open(out))["kernels"]["conv_wgrad_c8_kernel<C8Bf16, 5, false, 128, 128, false>"]
    assert "pass_mismatch" in w and "traffic_over_alg" not in w
