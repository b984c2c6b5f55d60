"""Assembles profiles/<tag>_end_of_round.md, <tag>_counters.md and <tag>_bench_line.json from the scratch output of
tools/profile_round.sh (gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>/) and a default bench line:
python tools/make_profile_docs.py <tag> <bench_line.json> [suite summary]     (run after tools/pmc_to_json.py)"""
import contextlib, io, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary

tag, bench_path = sys.argv[1], sys.argv[2]
suite = sys.argv[3] if len(sys.argv) > 3 else ""
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
pmc_dir = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
dst = os.path.join(ROOT, "profiles")
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()


def cap(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            fn(*a)
        except Exception as e:
            print(f"(not available: {e})")
    return buf.getvalue()


def read(name, base=src):
    p = os.path.join(base, name)
    return open(p).read() if os.path.exists(p) else f"(missing: {name})\n"


def clean(text):
    return "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l and "simple_timer" not in l and "rocprofv3" not in l
                     and "generateRocpd" not in l and "tool.cpp" not in l) + "\n"


raw = [l for l in open(bench_path).read().strip().splitlines() if l.startswith("{")]
line = raw[-1]
d = json.loads(line)
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")
rf = d.get("roofline", {})
c5 = (d.get("config5") or {}).get("wide_lookahead", {})
md = [f"# {tag} — end of round: rocprofv3 evidence (MI355X, commit {head})\n",
      "Collected by `tools/profile_round.sh` + the default `python bench.py` in one `gpurun` call" + (f" ({suite})" if suite else "") +
      "; raw rocpd databases stay in `gpurun_out/` (scratch).\n",
      "## default bench line (`python bench.py`)\n```\n" + line + "\n```\n",
      f"Headline: **{d['value'] / 1e9:.2f} G arcs/s, {d['ms_per_step']} ms per step**; `shortest_path(T)` alone {d.get('ms_shortest_path_T')} ms, fused batch "
      f"alone {d.get('ms_compose_shortest_path_batch')} ms.  `roofline`: {rf.get('launches')} launches, {rf.get('solve_relax_kernel_ms')} ms of relaxation "
      f"kernels per solve (HIP events, un-profiled) -> 212 MB / that = {rf.get('achieved')} GB/s = **{rf.get('frac')}** of 8 TB/s; traffic "
      f"{rf.get('traffic_over_algorithmic')} x algorithmic (stale: {rf.get('traffic_stale')}).  `roofline_vs_size`: "
      + " / ".join(f"{p.get('frac')}" for p in (d.get('roofline_vs_size') or {}).get('points', [])) + " at "
      + " / ".join(f"{p['states'] // 1000000}M" for p in (d.get('roofline_vs_size') or {}).get('points', [])) + " states.  "
      f"`step_512_acceptors`: {json.dumps(d.get('step_512_acceptors'))}.  `cold_query_ms`: {json.dumps((d.get('cold_query_ms') or {}).get('fresh_handle_warm_process'))}.  "
      f"`config5.wide_lookahead`: first {c5.get('first_call_ms')} ms, second {c5.get('second_call_ms')}, then {c5.get('repeated_calls_ms')} "
      f"(spread {c5.get('spread')}), `roofline_compose.frac` {(c5.get('roofline_compose') or {}).get('frac')}.\n",
      "## The two clocks of the relaxation: HIP events and the profiler, SAME process\n",
      "`tools/sp_repeat.py 1000000 12` under `rocprofv3 --kernel-trace` (first block) and without the profiler (second): the tool prints the HIP-event "
      "bracket around the pre-queued launch chain of its own solves.\n```\n" + clean(read("sp_alone.log")) + "---- without the profiler\n" +
      clean(read("sp_alone_unprofiled.log")) + "```\n",
      "Kernel durations of the profiled process (sum per solve below) against its own event bracket: the profiler's per-dispatch completion handling "
      "stretches a chain of back-to-back launches; the bench line's figure is the un-profiled bracket.\n",
      "## kernel table: the bench step (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras`)\n",
      cap(rocpd_summary.trace, os.path.join(src, "trace_results.db")),
      "\n## the relaxation alone at 1M states (`tools/sp_repeat.py 1000000 12`)\n", cap(rocpd_summary.trace, os.path.join(src, "sp_alone_results.db")),
      "\n## 2M states (8192-state blocks, every launch resident)\n", cap(rocpd_summary.trace, os.path.join(src, "sp_2m_results.db"), 424e6),
      "\n## 5M states (atomic sweeps: the default there)\n", clean(read("sp_5m.log")), cap(rocpd_summary.trace, os.path.join(src, "sp_5m_results.db"), 1060e6),
      "\n## the wide look-ahead driver (`tools/wide_lookahead_run.py 5000000 4`)\n```\n" + clean(read("wide.log")) + "```\n",
      cap(rocpd_summary.trace, os.path.join(src, "wide_results.db")),
      "\n## level stamps of the resident launch (`WFST_SSSP_RES_TRACE`, `tools/res_levels.py`: microseconds from the launch's start)\n```\n" + read("res_levels.txt") + "```\n"]
open(os.path.join(dst, f"{tag}_end_of_round.md"), "w").write("\n".join(md))
cm = [f"# {tag} — PMC passes of the relaxation at 1M states (commit {head})\n",
      "`tools/profile_round.sh`: every pass runs `tools/sp_repeat.py 1000000 12` under `rocprofv3 --pmc <one group> --kernel-trace`; sums are PER SOLVE.  "
      "`profiles/pmc_relax_traffic.json` (read by bench.py) is written from the same databases by `tools/pmc_to_json.py`.\n```\n" + read("pmc_summary.txt") + "```\n"]
open(os.path.join(dst, f"{tag}_counters.md"), "w").write("\n".join(cm))
print("wrote", f"{tag}_end_of_round.md", f"{tag}_counters.md", f"{tag}_bench_line.json")
