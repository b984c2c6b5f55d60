"""Assembles profiles/<tag>_*.md from the scratch output of tools/profile_round3.sh (gpurun_out/prof_<tag>/ and
gpurun_out/pmc_<tag>_mbox): python tools/make_profile_docs_r3.py <tag>      (run after tools/pmc_to_json.py)"""
import collections, contextlib, io, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
pmc_dir = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_mbox")
dst = os.path.join(ROOT, "profiles")
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()


def cap(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            fn(*a)
        except Exception as e:  # a missing database must not lose the rest of the set
            print(f"(not available: {e})")
    return buf.getvalue()


def read(name, base=src):
    p = os.path.join(base, name)
    return open(p).read() if os.path.exists(p) else f"(missing: {name})\n"


def clean(text):
    return "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l) + "\n"


def per_launch_instructions(db):
    """SQ counters of one solve, launch by launch, per wave (the SQ block samples a subset of the waves)."""
    c = sqlite3.connect(db)
    per = collections.OrderedDict()
    for d, n, cn, v in c.execute("select dispatch_id, name, counter_name, counter_value from pmc_events order by dispatch_id"):
        per.setdefault(d, {"name": n})[cn] = v
    ids = list(per)
    starts = [i for i, d in enumerate(ids) if "setup_kernel" in per[d]["name"]]
    if len(starts) < 2:
        print("(fewer than two solves in the pass)")
        return
    print("| # | kernel | waves sampled | VALU / wave | SALU / wave | LDS / wave | VMEM / wave | wave cycles / wave (x4) |")
    print("|---:|---|---:|---:|---:|---:|---:|---:|")
    for k, d in enumerate(ids[starts[-2]:starts[-1]]):
        p = per[d]
        w = p.get("SQ_WAVES", 0) or 1
        nm = rocpd_summary.short(p["name"]).split("(")[0].replace("void ", "")
        print(f"| {k} | `{nm}` | {w:.0f} | {p.get('SQ_INSTS_VALU', 0) / w:.0f} | {p.get('SQ_INSTS_SALU', 0) / w:.0f} | "
              f"{p.get('SQ_INSTS_LDS', 0) / w:.0f} | {p.get('SQ_INSTS_VMEM', 0) / w:.0f} | {p.get('SQ_WAVE_CYCLES', 0) / w:.0f} |")


raw = [l for l in read("bench_line.json").strip().splitlines() if l.startswith("{")]
line = raw[-1] if raw else "{}"
d = json.loads(line)
if raw:
    open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")
rf = d.get("roofline", {})
md = [f"# {tag} — end of round 3: rocprofv3 evidence (MI355X, commit {head})\n",
      "All of it collected by `tools/profile_round3.sh` in one `gpurun` call; raw rocpd databases stay in `gpurun_out/` (scratch).\n",
      "## default bench line (`python bench.py`)\n```\n" + line + "\n```\n"]
if raw:
    md.append(f"Headline: {d['value'] / 1e9:.2f} G arcs/s, {d['ms_per_step']} ms per step; `shortest_path(T)` alone {d.get('ms_shortest_path_T')} ms, "
              f"fused batch alone {d.get('ms_compose_shortest_path_batch')} ms.  `roofline`: {rf.get('launches')} launches, "
              f"{rf.get('solve_relax_kernel_ms')} ms of `{rf.get('kernel')}` per solve -> 212 MB / that = {rf.get('achieved')} GB/s = "
              f"**{rf.get('frac')}** of 8 TB/s (SURVEY §8(d) accounting); traffic per launch {rf.get('traffic')} B.\n")
md += ["## kernel table: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras` (both requests overlapped)\n",
       cap(rocpd_summary.trace, os.path.join(src, "trace_results.db")),
       "\n## the relaxation alone: `rocprofv3 --kernel-trace -- python tools/sp_repeat.py 1000000 12` (un-profiled solves, nothing else on the GPU)\n",
       cap(rocpd_summary.trace, os.path.join(src, "sp_alone_results.db")),
       "\n### timeline of one of those solves (launch order; duration and idle gap before each kernel)\n",
       "Launch 0 of `sssp_mbox_kernel` is the head (NARROW, one workgroup at work), the last two are the COLLECT and the NARROW launch that drains the search; the others are WIDE sweeps.\n",
       cap(rocpd_summary.timeline, os.path.join(src, "sp_alone_results.db"), -3),
       "\n## 5M states / 50M arcs (config 5's size): `tools/sp_repeat.py 5000000 6` under both relaxation kernels\n",
       "### `WFST_SSSP_MAILBOX=1` (`sssp_mbox_kernel<true>`, 1221 blocks)\n", clean(read("sp5m_mode1.log")),
       cap(rocpd_summary.trace, os.path.join(src, "sp5m_mode1_results.db"), 1060e6),
       "\n### `WFST_SSSP_MAILBOX=0` (`sssp_relax_kernel`, the default beyond 768 blocks)\n", clean(read("sp5m_mode0.log")),
       cap(rocpd_summary.trace, os.path.join(src, "sp5m_mode0_results.db"), 1060e6),
       "\n## where a sweep's time goes: phase stamps of `sssp_mbox_kernel` (`WFST_SSSP_MBOX_TRACE`, `tools/mbox_phases.py`)\n",
       "Per sweep: blocks that did something / slept, states expanded, messages sent, then for every phase boundary the time since the "
       "first block of the sweep started at which the LAST (median) busy block passed it.  The stamps themselves cost time (a wall-clock "
       "read and a store by the first wave at every boundary): stamped sweeps are ~2 us longer than the same sweeps in the timeline above.\n```\n",
       read("mbox_phases.txt"), "```\n",
       "\n## atomic sweeps vs mailbox sweeps, per launch (profiled solves: one launch at a time, events around it)\n```\n", clean(read("sweep_compare.txt")), "```\n"]
open(os.path.join(dst, f"{tag}_end_of_round.md"), "w").write("\n".join(md))

md = [f"# {tag} — PMC counters of the relaxation (rocprofv3 --pmc, one counter group per pass; commit {head})\n",
      "`tools/pmc_relax.sh`: every pass runs `tools/sp_repeat.py 1000000 6` (six un-profiled `shortest_path(T)` solves on the C3 graph); "
      "sums are PER SOLVE.  SQ_* counters come from a subset of the shader engines (SQ_WAVES per launch says how many waves were seen); "
      "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  FETCH_SIZE / WRITE_SIZE are KB; "
      "`profiles/pmc_relax_traffic.json` is what `bench.py` reads for `roofline.traffic`.\n",
      "```\n", clean(read("pmc_mbox.txt")), "```\n",
      "## instructions per wave, launch by launch (one solve of the `sq1` pass)\n",
      "A wave of a nearly idle WIDE sweep executes ~550 vector + ~450 scalar + ~45 LDS instructions and lives ~23 k cycles: with four "
      "waves per SIMD the vector instructions alone are ~9 k cycles of issue, the rest is waiting on the three dependent memory trips "
      "and on LDS round trips (LAB_NOTEBOOK.md, round 3).\n",
      cap(per_launch_instructions, os.path.join(pmc_dir, "sq1_results.db"))]
open(os.path.join(dst, f"{tag}_counters.md"), "w").write("\n".join(md))

# the BASELINE configs beside the headline: what the bench line's extras measured
if raw:
    md = [f"# {tag} — configs[4] (look-ahead + n-best at 5M states), the batch-size sweep and configs[1] (commit {head})\n",
          "From the `config5`, `batch_sweep` and `config2_single_string` keys of the bench line of this set (`bench.py` extras: untimed, one-off).\n"]
    for k in ("config5", "batch_sweep", "config2_single_string", "cold_query_ms", "reference_harness_split"):
        if k in d:
            md += [f"## `{k}`\n```\n" + json.dumps(d[k], indent=1) + "\n```\n"]
    open(os.path.join(dst, f"{tag}_config5_batch_sweep.md"), "w").write("\n".join(md))
print("written")
