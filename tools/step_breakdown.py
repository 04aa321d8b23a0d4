#!/usr/bin/env python3
"""Whole-step view of the committed rocprofv3 kernel statistics (profiles/r03_bench_kernel_stats.csv, 30 profiled steps of the
8192-graph ZINC-shape NGNN step): per kernel family the time per step, the ALGORITHMIC bytes its tuple-level launches move
(rows once, indices once), the rate, and the floor the box's measured direction rates give for that read / write mix
(profiles/r03_read_bw_probe.json: streamed reads 6.95 TB/s, rows gathered by index 5.8 TB/s, writes 5.4 TB/s in 32-KB pieces /
4.5 TB/s in 4-16-KB pieces, reads and writes do not overlap).  usage: step_breakdown.py [kernel_stats.csv] [steps]"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_bench_kernel_stats.csv")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30

T, E, M, D = 1_775_544, 405_872, 3_508_784, 128           # tuples, edges, message triples, hidden (bench.py's batch)
NODES = 188_744
row = D * 2                                               # bf16 row bytes
S = T * row                                               # one tuple-level (nnz, d) stream: 455 MB
RD_STREAM, RD_GATHER, WR_BIG, WR_SMALL = 6.95e12, 5.8e12, 5.4e12, 4.5e12

# family -> (substring of the kernel name, tuple-level launches per step, gathered-read bytes, streamed-read bytes, written bytes, write rate)
FAM = [
    ("spspmm forward + residual, by-tuple backward (seg_gmr_fast<bf16,SUM,BOTH>)", "seg_gmr_fast_kernel<pygho::bf16, 0, 0, false, true, false, false, 0>", 12,
     S, (6 * (S + 8 * M + 4 * T) + 6 * (8 * M + 4 * T)) / 12, S, WR_SMALL),      # bench.py's accounting: the edge operand is read through its 16-row table
    ("spspmm by-edge backward (seg_gmr_window)", "seg_gmr_window_kernel<pygho::bf16", 6, 2 * S, 8 * M + 4 * E + (5 / 6) * E * row, E * row, WR_SMALL),
    ("dense backward: BN/act backward + dX + dW, pre-activation recomputed (bn_bwd_linear_dw)", "bn_bwd_linear_dw_kernel<pygho::bf16, 128, 2, true>", 6, 0, 3 * S, S, WR_BIG),
    ("backward channel sums, pre-activation recomputed (rowblock_linear<2,2>)", "rowblock_linear_kernel<pygho::bf16, 128, 2, 2>", 6, 0, 2 * S, 0, WR_BIG),
    ("Linear + BatchNorm + SiLU (rowblock_linear<1,2>)", "rowblock_linear_kernel<pygho::bf16, 128, 1, 2>", 6, 0, S, S, WR_BIG),
    ("BatchNorm statistics of the Linear's output (rowblock_linear<0,0>)", "rowblock_linear_kernel<pygho::bf16, 128, 0, 0>", 6, 0, S, 0, WR_BIG),
    ("tuple initialisation forward (unit_triple)", "unit_triple_kernel<pygho::bf16", 1, 0, 12 * T, S, WR_SMALL),
    ("tuple initialisation backward in one pass on symmetric tuple sets (pair_bwd)", "pair_bwd_kernel<pygho::bf16>", 1, S, S + 12 * T, 2 * NODES * row, WR_SMALL),
    ("tuple initialisation backward, by row / by column (seg_gmr_fast, three operands)", "seg_gmr_fast_kernel<pygho::bf16, 0, 0, false, true, false, true, 0>", 2, S / 2, S / 2 + 12 * T, NODES * row, WR_SMALL),
    ("tuple initialisation backward, feature table (seg_gmr_fast, three operands, f32 partials)", "seg_gmr_fast_kernel<pygho::bf16, 0, 0, false, true, true, true, 0>", 1, S, 12 * T, 0, WR_SMALL),
    ("subgraph mean pooling (seg_gmr_fast<bf16,MEAN,LHS>)", "seg_gmr_fast_kernel<pygho::bf16, 1, 1, false, true, false, false, 0>", 1, 0, S + 4 * NODES, NODES * row, WR_SMALL),
]

rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"kernel time per step: {total / 1e3:.2f} ms ({path.split('/')[-1]}, {steps} steps)\n")
print("| kernels | us per step | launches per step | algorithmic GB per tuple-level launch | TB/s on those | floor by the box's direction rates | launch / floor |")
print("|---|---|---|---|---|---|---|")
seen = 0.0
for name, key, n_big, rd_g, rd_s, wr, wr_rate in FAM:
    hit = [r for r in rows if key in r["Name"]]
    if not hit:
        continue
    r = hit[0]
    per_step = float(r["TotalDurationNs"]) / steps / 1e3
    calls = int(r["Calls"]) / steps
    seen += per_step
    # the node-level launches of the same kernels (2 of 8 for the block kernels) move 1/9.4 of the bytes: scale them out
    small = calls - n_big
    frac_small = small * (NODES / T) / (n_big + small * (NODES / T)) if small > 0 else 0.0     # node-level launches: 1 / 9.4 of the rows
    per_big = per_step * (1 - frac_small) / n_big
    gb = (rd_g + rd_s + wr) / 1e9
    floor_us = (rd_g / RD_GATHER + rd_s / RD_STREAM + wr / wr_rate) * 1e6
    print(f"| {name} | {per_step:.0f} | {calls:.0f} | {gb:.2f} | {gb / per_big * 1e3:.2f} | {floor_us:.0f} us | {per_big / floor_us:.2f} |")
print(f"| everything else (≈ 120 launches below 50 us: gathers, folds, finalisations, casts, library GEMMs, optimizer) | {total - seen:.0f} | | | | | |")
