#!/usr/bin/env python3
"""The numbers DESIGN.md section 5 / 6 and README.md quote, from one round's collected files (collect_round.sh <tag>).
usage: python profiles/summarize_round.py <dir> <tag>"""
import json, os, sys
d, tag = sys.argv[1], sys.argv[2]


def load(name):
    try:
        return json.load(open(os.path.join(d, f"{tag}_{name}")))
    except (OSError, ValueError) as e:
        print(f"({name}: {e})")
        return None


def row(name, value, ms, r, extra=""):
    w = r.get("wave_cycles", {})
    by = r.get("frac_by_rule")
    if by:
        print(f"    frac_by_rule: valu {by['valu']}  hbm_counter {by['hbm_counter']}  hbm_counter_upper {by['hbm_counter_upper']}  survey_8d_algorithmic {by['survey_8d_algorithmic']}; "
              f"launches {[(b['frames'], b['waves_per_simd'], b['kernel_ms']) for b in r.get('launches', [])]}; requests {r.get('fabric_read_requests')}")
    print(f"| {name} | {value:,.0f} | {ms:.2f} | {r['frac']} = {r.get('valu_issue_frac')} · {r.get('lane_utilisation')} | {r['hbm']['achieved']:,.0f} → {r['hbm']['frac']} "
          f"({r['hbm']['frac_of_achievable']}) | {r.get('l2', {}).get('achieved', 0):,.0f}, {r.get('l2', {}).get('hit_rate')} | "
          f"{w.get('issuing')} / {w.get('waiting_for_memory')} / {w.get('issue_stalled')} | {r['box_tests_per_ray']} + {r['tri_tests_per_ray']} |{extra}")


for f, label in (("dragon_1080p_bench_driver_args.json", "dragon-class 870 k tris (headline; --steps 20 --warmup 5)"), ("dragon_1080p_bench.json", "same, default arguments (256 spp)"),
                 ("demo_1080p_bench.json", "demo 1 998 tris (--workload demo)")):
    j = load(f)
    if not j:
        continue
    r = j["roofline"]
    row(label, j["value"], j["ms_per_step"], r)
    print(f"    kernel_ms {r['kernel_ms']} excl {r['kernel_ms_exclusive']} all {r['kernel_ms_all_launches']} over {r['launches_all']}; traffic {r['traffic'] / 1e9:.1f} GB/launch; "
          f"alg {r['algorithmic_GBps']} GB/s, traffic/alg {r['traffic_over_algorithmic']}; bytes/ray {r['bytes_per_ray']}; VALU insts/launch {r.get('valu_insts_per_launch', 0) / 1e9:.2f} G; "
          f"VMEM {r.get('pmc_counters', {}).get('SQ_INSTS_VMEM_RD')}")
    for k, v in (j.get("also") or {}).items():
        print(f"    also.{k}: {v['value']:,.0f} Mrays/s" + (f", {v.get('box_tests_per_ray')} + {v.get('tri_tests_per_ray')} per ray" if "box_tests_per_ray" in v else "") + (f", {v.get('ms_per_frame')} ms per frame" if "ms_per_frame" in v else ""))
    if j.get("forest"):
        fr = j["forest"]
        row("forest 10 M tris, 2.08 GB (forest leg, 1080p)", fr["value"], fr["ms_per_step"], fr["roofline"])
        print(f"    forest kernel_ms {fr['roofline']['kernel_ms']} excl {fr['roofline']['kernel_ms_exclusive']}; traffic {fr['roofline']['traffic'] / 1e9:.0f} GB/launch")
    if j.get("cpu_baseline"):
        c = j["cpu_baseline"]
        print(f"    cpu_baseline: {c['value']} Mrays/s on {c['cores']} threads; scaling {c['thread_scaling_Mrays_per_s']}")
for f in ("bench_under_rocprof_default.json", "bench_under_rocprof_steps20warmup5.json"):
    j = load(f)
    if j:
        print(f"{f}: value {j['value']}, kernel_ms_all_launches {j['roofline']['kernel_ms_all_launches']} over {j['roofline']['launches_all']}")
for f in ("dragon_1080p_kernel_stats_default.csv", "dragon_1080p_kernel_stats_steps20warmup5.csv"):
    try:
        for line in open(os.path.join(d, f"{tag}_{f}")):
            if "k_raytrace_sm" in line:
                p = line.strip().split(",")
                print(f"{f}: calls {p[-6]} avg ns {p[-4]}")
    except OSError as e:
        print(f"({f}: {e})")
for f in ("scaling_model_steps20.log", "scaling_model_steps16.log", "scaling_model_steps4.log", "node_render_loop.json", "spf_same_frames.log", "interactive.log", "rank_timeline.log", "fullscreen_time.log"):
    try:
        print(f"--- {f}\n" + open(os.path.join(d, f"{tag}_{f}")).read()[:2500])
    except OSError as e:
        print(f"({f}: {e})")
