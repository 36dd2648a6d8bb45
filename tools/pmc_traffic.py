#!/usr/bin/env python
"""HBM-side traffic of the bench's dominant kernel from the PMC counters, with its own calibration (GPU box).

Runs, each in its own rocprofv3 pass (PMC only with --kernel-trace, as gpurun requires):
  1. bench.py under --pmc FETCH_SIZE, then under --pmc WRITE_SIZE           -> counters per launch of every kernel
  2. a streaming float4 copy of KNOWN size (lins_debug_stream_copy) under the same two counters
     -> calibration factors  known bytes / counter bytes  for 16-B-per-lane coalesced reads and writes
and writes profiles/<tag>_pmc_traffic.json, stamped with the content digest of the sources the library was built
from (bench.py only reports a record whose stamp matches the sources it runs).

What the numbers mean (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count the L2's fabric-side
requests, Infinity-Cache hits included — an UPPER bound of HBM traffic; on gfx950 FETCH_SIZE tallies wide
coalesced reads at half their bytes.  bytes_lo takes the counters as they are, bytes_hi applies the factors
calibrated on the copy (reads x ~2); the kernel's dword scratch traffic is not the calibrated pattern, so the
truth lies in [lo, hi].
usage: tools/pmc_traffic.py <tag> [bench args...]"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def counters(cmd, counter, tag):
    """One rocprofv3 pass (PMC + kernel trace only).  counter: a name -> {kernel: (sum, dispatches)};
    a list of names -> {kernel: {name: sum / dispatches}}."""
    names = [counter] if isinstance(counter, str) else list(counter)
    d = os.path.join(OUT, f"_pmc_{tag}_{names[0]}")
    subprocess.run(["rm", "-rf", d])
    # (counters are summed per LAUNCH: the runs under the counters pin the one-launch form of a step — lins_set_launch_queues 1 —
    # so that "per launch" is "per step of the whole batch", as in every earlier round's record)
    env = dict(os.environ, TMPDIR="/tmp", LINS_ENABLE_DEBUG_KNOBS="1", LINS_SPLIT_STREAMS="0")
    # (bounded: bench.py runs these passes inside the driver's run — a profiler that hangs must cost a fallback to the
    # committed record, not the bench line; the pass runs in a process group of its own so that the limit ends all of it)
    proc = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc"] + names + ["-d", d, "--"] + cmd, cwd="/tmp", env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=int(os.environ.get("LINS_PMC_TIMEOUT", "180")))
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)  # (the group this call created: rocprofv3 and the run under it)
        except OSError:
            pass
        proc.communicate()
        subprocess.run(["rm", "-rf", d])
        raise RuntimeError(f"rocprofv3 pass {names[0]} did not finish in time")

    class _P:
        stdout = out
    p = _P()
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    res = {}
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        for nm in names:
            for name, val, n, rows in cur.execute(
                    "select k.name, sum(p.counter_value), count(distinct k.dispatch_id), count(*) from pmc_events p join kernels k"
                    " on p.dispatch_id = k.dispatch_id where p.counter_name = ? group by k.name", (nm,)):
                if isinstance(counter, str):
                    res[name] = (val, n)
                elif nm.startswith("GRBM_"):  # one row per XCD, each the whole launch: the mean, not the sum
                    res.setdefault(name, {})[nm] = val / rows
                else:                         # SQ: one row per shader engine: the sum is the device total
                    res.setdefault(name, {})[nm] = val / n
    subprocess.run(["rm", "-rf", d])
    return res, p.stdout.decode()[-400:]


def measure(tag, bench_args, with_valu=True):
    """The record (dict) of one measurement: FETCH_SIZE / WRITE_SIZE passes over bench.py and over the calibration copy,
    then (with_valu) one SQ pass.  bench.py calls this itself when rocprofv3 is on the box (traffic_source "live")."""
    import __graft_entry__ as g

    os.environ["LINS_BENCH_NO_LIVE_PMC"] = "1"  # (the bench runs under the counters must not start counter runs of their own)

    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"] + bench_args
    copy = [sys.executable, "-c",
            "import sys, ctypes as C, importlib; sys.path.insert(0, %r);"
            "pkg = importlib.import_module('lins---lidar-inertial-slam_amd'); ieskf = importlib.import_module('lins---lidar-inertial-slam_amd.ieskf');"
            "c = ieskf.IeskfContext(pkg.default_params(), max_batch=1024, max_targets=8192); L = ieskf.lib();"
            "L.lins_debug_stream_copy.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]; g = C.c_double(0);"
            "assert L.lins_debug_stream_copy(c._h, 1 << 28, 3, C.byref(g)) == 0; print('copy GB/s', g.value)" % ROOT]
    rec = {"tag": tag, "source_digest": g._source_digest(), "command": " ".join(bench[1:]), "kernels": {}}
    search = "auto"
    if "--search" in bench_args:
        search = bench_args[bench_args.index("--search") + 1]
    rec["search"] = search
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        res, tail = counters(bench, counter, tag)
        if not res:
            print("no counters from", counter, tail, file=sys.stderr)
        for name, (val, n) in res.items():
            rec["kernels"].setdefault(name, {})[counter + "_KB_per_launch"] = val / n
        cres, ctail = counters(copy, counter, tag + "cal")
        for name, (val, n) in cres.items():
            if "copy" in name:
                cal[counter] = {"kernel": name, "counter_KB_per_launch": val / n, "known_KB_per_launch": (1 << 28) / 1024.0,
                                "factor": ((1 << 28) / 1024.0) / (val / n) if val else None}
    rec["calibration"] = cal
    # the dominant kernel = the one with the most traffic among the IESKF kernels
    iesk = {k: v for k, v in rec["kernels"].items() if "ieskf" in k and "joseph" not in k}
    if iesk:
        dom = max(iesk, key=lambda k: sum(iesk[k].values()))
        f = iesk[dom].get("FETCH_SIZE_KB_per_launch", 0.0) * 1024
        w = iesk[dom].get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
        ff = (cal.get("FETCH_SIZE") or {}).get("factor") or 2.0
        wf = (cal.get("WRITE_SIZE") or {}).get("factor") or 1.0
        rec.update(kernel=dom, fetch_size_bytes=f, write_size_bytes=w, bytes_lo=f + w, bytes_hi=ff * f + max(wf, 1.0) * w,
                   meaning="per launch; fabric-side (L2 <-> Infinity Fabric) bytes, Infinity-Cache hits included: an upper bound "
                           "of HBM traffic; lo = counters as reported, hi = with the factors calibrated on a streaming copy")
    # VALU side of the same kernel (SURVEY.md section 8d asks for both fractions): SQ counters are quad-cycles summed
    # over all SIMDs; GRBM_GUI_ACTIVE is the launch in shader cycles
    sq, tail = ({}, "") if not with_valu else counters(bench, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"], tag + "sq")
    if iesk and rec.get("kernel") in sq:
        c = sq[rec["kernel"]]
        n_simd = 256 * 4
        act, thr = c.get("SQ_ACTIVE_INST_VALU", 0.0), c.get("SQ_THREAD_CYCLES_VALU", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        rec["valu"] = {"insts_valu_per_launch": c.get("SQ_INSTS_VALU"), "active_inst_valu_quadcycles": act,
                       "thread_cycles_valu": thr, "lanes_per_inst": (thr / act) if act else None,
                       "gui_active_cycles": gui,
                       "busy_frac": (act / (n_simd * gui / 4.0)) if gui else None,
                       "wave_wait_frac": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                       "meaning": "per launch of the dominant kernel; busy_frac = VALU-issuing quad-cycles summed over the SIMDs / (1024 SIMDs x "
                                  "launch quad-cycles); lanes_per_inst = thread-cycles / instruction-cycles of VALU work (of 64)"}
    else:
        print("no SQ counters:", tail, file=sys.stderr)
    # LDS side (bank-conflict cycles per active LDS cycle, next to round 5's 0.85) and the scalar instruction count
    if with_valu and rec.get("valu") is not None:
        lds, tail = counters(bench, ["SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_SALU"], tag + "lds")
        c = lds.get(rec["kernel"]) or {}
        if c.get("SQ_ACTIVE_INST_LDS"):
            rec["valu"].update(lds_bank_conflict_cycles=c.get("SQ_LDS_BANK_CONFLICT"), lds_active_cycles=c.get("SQ_ACTIVE_INST_LDS"),
                               lds_conflict_per_active=c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"],
                               insts_lds_per_launch=c.get("SQ_INSTS_LDS"), insts_salu_per_launch=c.get("SQ_INSTS_SALU"))
        else:
            print("no LDS counters:", tail, file=sys.stderr)
    return rec


def main():
    tag = sys.argv[1]
    rec = measure(tag, sys.argv[2:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json")
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
