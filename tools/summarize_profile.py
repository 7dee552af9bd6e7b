#!/usr/bin/env python3
"""Summarise rocprofv3 runs of bench.py into profiles/<tag>_summary.{md,json}.

  python tools/summarize_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [--workload NAME]

* <stats_dir>: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py`
* <fetch_dir>/<write_dir>: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes.
HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
stream, so fetch bytes are shown both raw and doubled ("x2") -- the truth lies
between for mixed-width access; WRITE_SIZE is uncalibrated.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("tmi::", "")
    return n


def read_stats(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append(dict(kernel=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                         avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"])))
    return rows, f


def read_counter(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: acc[k] / cnt[k] * 1024.0 for k in acc}  # bytes per launch


def main():
    workload = "venice1778_heavy"  # bench.py's default
    if "--workload" in sys.argv:
        i = sys.argv.index("--workload")
        workload = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    tag, stats_dir = sys.argv[1], sys.argv[2]
    rows, stats_file = read_stats(stats_dir)
    fetch = read_counter(sys.argv[3], "FETCH_SIZE") if len(sys.argv) > 3 else {}
    write = read_counter(sys.argv[4], "WRITE_SIZE") if len(sys.argv) > 4 else {}
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    for r in rows:
        r["fetch_bytes_raw"] = fetch.get(r["kernel"])
        r["write_bytes"] = write.get(r["kernel"])
        if r["fetch_bytes_raw"] is not None and r["write_bytes"] is not None:
            r["hbm_bytes_x2fetch"] = 2 * r["fetch_bytes_raw"] + r["write_bytes"]
            r["hbm_GBs_x2fetch"] = r["hbm_bytes_x2fetch"] / (r["avg_us"] * 1e-6) / 1e9
    json.dump(rows, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
    with open(os.path.join(out_dir, f"{tag}_summary.md"), "w") as fh:
        fh.write(f"# rocprofv3 summary `{tag}` (bench.py, workload {workload}, synthetic, 1 x MI355X)\n\n")
        fh.write("| kernel | calls | avg us | % time | FETCH_SIZE MB/launch (raw) | WRITE_SIZE MB/launch | HBM GB/s (2 x fetch + write) |\n")
        fh.write("|---|---|---|---|---|---|---|\n")
        for r in rows:
            f = "" if r["fetch_bytes_raw"] is None else f"{r['fetch_bytes_raw'] / 1e6:.1f}"
            w = "" if r["write_bytes"] is None else f"{r['write_bytes'] / 1e6:.1f}"
            g = "" if "hbm_GBs_x2fetch" not in r else f"{r['hbm_GBs_x2fetch']:.0f}"
            fh.write(f"| {r['kernel']} | {r['calls']} | {r['avg_us']:.1f} | {r['pct']:.2f} | {f} | {w} | {g} |\n")
    import shutil
    shutil.copy(stats_file, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    # per kernel CLASS traffic (bench.py's classes) for the roofline `traffic` field
    cls_of = {"linearize_kernel": "linearize", "point_eliminate_kernel": "point_eliminate",
              "camera_diag_kernel": "camera_diag", "schur_offdiag_kernel": "schur_offdiag",
              "precond_invert_kernel": "preconditioner", "spmv_rows_kernel": "spmv",
              "spmv_cols_kernel": "spmv", "back_substitute_kernel": "back_substitute",
              "implicit_tracks_kernel": "spmv_matrix_free", "implicit_cameras_kernel": "spmv_matrix_free",
              "implicit_tracks_q_kernel": "spmv_matrix_free", "implicit_cameras_q_kernel": "spmv_matrix_free",
              "mfc::product_kernel": "spmv_matrix_free", "mfc::reduce_kernel": "spmv_matrix_free",
              "cost_kernel": "update_cost", "update_points_kernel": "update_cost",
              "update_cameras_kernel": "update_cost", "schur_offdiag_aq_kernel": "schur_offdiag",
              "pcg_step_kernel": "pcg_vector", "pcg_p_kernel": "pcg_vector", "pcg_init_kernel": "pcg_vector",
              "camera_prepare_kernel": "linearize",
              # round 5 (direct_diag.h): the camera side and the trial cost view by view
              "ddg::camera_diag_direct_kernel": "camera_diag", "ddg::camera_diag_direct_reduce_kernel": "camera_diag",
              "ddg::cost_view_kernel": "update_cost", "pos_coef_kernel": "linearize"}
    classes = {}
    for r in rows:
        base = r["kernel"].split("<")[0]
        if base in cls_of and "hbm_bytes_x2fetch" in r:
            classes[cls_of[base]] = classes.get(cls_of[base], 0.0) + r["hbm_bytes_x2fetch"]
    sys.path.insert(0, os.path.dirname(out_dir))
    import bench  # engine_source_sha(): the stamp bench.py checks before quoting these bytes
    if fetch and write:  # only a run with both PMC passes replaces the traffic table
      # the default workload's table is pmc_latest.json; every other workload keeps its own (bench.py reads the
      # reference-default one for variants.reference_defaults.roofline.traffic)
      name = "pmc_latest.json" if workload == "venice1778_heavy" else f"pmc_{workload}.json"
      json.dump(dict(tag=tag, workload=workload, correction="2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes",
                     engine_source_sha=bench.engine_source_sha(), git_head=os.environ.get("TMI_GIT_HEAD"),
                     classes=classes), open(os.path.join(out_dir, name), "w"), indent=1)
    print(open(os.path.join(out_dir, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
