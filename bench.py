#!/usr/bin/env python3
"""bench.py — IF Msamples/s through the tracking correlators; x real-time @ 12-ch GPS L1 C/A.

Main line (BASELINE.json configs[1]): 12 GPS L1 C/A channels, 1-ms coherent E/P/L correlators, 60 s of
synthetic int8 I/Q IF at 18 Msps (2.16 GB resident in HBM), C/N0 45 dB-Hz, Doppler U(-5,5) kHz, seed 20241008+2.

  1. the record is synthesised in HBM (csrc/synth.hip);
  2. closed-loop tracking (gc_track: discriminators and loop filters on the host, the correlator in a persistent host-fed
     kernel; gc_track_device: the loop closed on the GPU) produces the per-epoch state -> x real-time (closed loop);
  3. a "step" = ONE batched replay pass of the hot path over the whole record: all channels x epochs blocks (720 000) in one
     launch, descriptors and IF resident in HBM.  `value` = IF samples through the correlators per second with every channel
     active (channel-samples / channels), whole job, all ranks.

With --config all (default at N = 1) the same line carries BASELINE.json's other configs under "configs": Galileo E1-C
CBOC(6,1,1/11) x 8, GPS L5 + BDS B2a x 16 at 50 Msps (int8 and int16 records), and one GPU's 8-channel share of the
all-constellation mix.  --config mix runs the whole 64-channel mix (configs[4]) sharded over the ranks by IF record (band).

N > 1: one rank per GPU.  `python bench.py --gpus N` starts the N ranks itself; under torch.distributed.run it joins the
launcher's ranks.  Channels shard across GPUs (tracking.m:133: channels share nothing but the read-only record), every rank
tracks its own 12 channels, so scaling is "weak".  The one exchange step of the sharded path is the record hand-over: the rank
that holds the IF record (rank 0; in the mix the first rank of each band) broadcasts it to the ranks that track other channels of
it - sharding.broadcast_record / distribute_band_records, RCCL over xGMI (process group "cpu:gloo,cuda:nccl"), adopted by the
engines without a copy (gc_attach_if) - and it is timed and reported (`handover`) next to the tracking numbers; --no-handover
lets every rank synthesise its own copy instead (A/B).  Acquisition shards by PRN (sharding.shard_prns / merge_acq_results).
The control plane (barrier + max over ranks) uses the same group's gloo side.  GC_BENCH_DEVICE=i puts every rank on device i
(functional check of the N > 1 path on a 1-GPU box): RCCL refuses two ranks on one GPU, so there the record travels through the
hosts (gloo) and the ranks take turns for their closed loops (persistent kernels need their whole grid resident).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "IF Msamples/s through tracking correlators; x real-time @ 12-ch GPS L1 C/A"


def _error_line(n: int, steps: int, warmup: int, error: str, **extra) -> str:
    """The ONE JSON line of a run that did not finish: same keys as a result line, value null, `error` says why."""
    if "stderr_tail" in extra:
        extra["stderr_tail"] = [str(ln)[-200:] for ln in extra["stderr_tail"][-8:]]
    return json.dumps({"metric": METRIC, "value": None, "unit": "IF Msamples/s", "n_gpus": n, "steps": steps, "warmup": warmup, "ms_per_step": None,
                       "higher_is_better": True, "error": str(error)[:800], **extra})


# =======================================================================================================================
# The printed line.  The driver keeps a bounded tail of stdout and parses the last JSON line in it: the line is the headline, its
# roofline and CPU baseline and a FLAT summary of the other legs (well under 4 KB); everything else goes to the detail file.
LINE_LIMIT = 4096
DETAIL_DEFAULT = os.path.join(ROOT, "bench_detail.json")

_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_CONFIG_KEYS = ("workload", "channels_per_gpu", "epochs", "blocks_per_step", "prewarm_ms", "parallelism")
_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "algorithmic_bytes_per_launch")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_line(result: dict, detail_path: str | None = None) -> str:
    """ONE JSON line < LINE_LIMIT bytes from the full result dict of run_l1ca (+ configs, acquisition, sweep, CPU leg) or run_mix."""
    out = {k: result[k] for k in _LINE_KEYS if k in result}
    out["dtype"] = _short(out.get("dtype", ""), 96)
    cfg = result.get("config", {})
    out["config"] = {k: (_short(cfg[k], 200) if isinstance(cfg[k], str) else cfg[k]) for k in _CONFIG_KEYS if k in cfg}
    roof = result.get("roofline")
    if roof:
        out["roofline"] = {k: (_short(roof[k], 100) if isinstance(roof[k], str) else roof[k]) for k in _ROOFLINE_KEYS if k in roof}
    cpu = result.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": _short(cpu.get("sample", ""), 260),
                               "host_cpu": (cpu.get("host_cpu") or {}).get("model"), "x_realtime": cpu.get("x_realtime")}
    for k in ("x_realtime_replay", "corr_msps", "replay_vs_closed_loop_max_dev", "channels", "channels_locked", "channels_locked_device_loop"):
        if k in result:
            out[k] = result[k]
    for k in ("closed_loop", "closed_loop_device", "closed_loop_host"):
        if isinstance(result.get(k), dict) and "x_realtime" in result[k]:
            out[k + "_x_realtime"] = result[k]["x_realtime"]
    if isinstance(result.get("handover"), dict):
        h = result["handover"]
        out["handover"] = {k: h[k] for k in ("backend", "bytes", "seconds") if k in h}
    if "rank_diag" in result:      # per rank: device, devices its process saw, how the RCCL probe went
        out["rank_diag"] = [{"r": d.get("rank"), "dev": d.get("device"), "seen": d.get("n_devices_seen"), "rccl": _short(d.get("rccl_probe"), 48)} for d in result["rank_diag"]]
    if "ranks" in result:
        out["ranks_locked"] = [r.get("channels_locked", sum(j.get("channels_locked", 0) for j in r.get("jobs", []))) for r in result["ranks"]]
    cfgs = result.get("configs")
    if cfgs:
        names = {"galileo_e1c_cboc_x8": "c3", "l5_b2a_x16_50msps_int8": "c4_int8", "l5_b2a_x16_50msps_int16": "c4_int16", "mix_share_l1_band_x8": "c5_share"}
        out["configs_frac"] = {names.get(k, k): v["replay"]["roofline"]["frac"] for k, v in cfgs.items()}
        out["configs_if_msps"] = {names.get(k, k): v["replay"]["if_msps"] for k, v in cfgs.items()}
        if "galileo_e1c_cboc_x8" in cfgs:
            out["c3_parity"] = "oracle-only (the CBOC(6,1,1/11) replica is not in the reference: GAL_E1C tracks BOC(1,1); modelled on BDS/B1C WB_tracking.m)"
    acq = result.get("acquisition")
    if acq:
        out["acq_l1ca_ms"] = round(acq["seconds"] * 1e3, 3)
        out["acq_all_scene_prns_found"] = bool(acq.get("all_scene_prns_found"))
        pk = acq.get("packages")
        if pk:
            out["acq_ms"] = {k: v["ms"] for k, v in pk.items()}
            out["acq_all_equal_to_reference"] = all(all(v["equal_to_the_references_acquisition_m"].values()) for v in pk.values())
            devs = [v["float64_guard"]["max_dev"] for v in pk.values() if v.get("float64_guard")]
            if devs:                       # float32 search values against the float64 re-evaluation of the same cells, worst package
                out["acq_f32_vs_f64_peak_max_rel"] = float("%.3g" % max(devs))
                out["acq_guard_ties"] = sum(v["float64_guard"]["ties"] for v in pk.values() if v.get("float64_guard"))
            l1 = pk.get("GPS_L1CA", {}).get("roofline", {}).get("compute")
            if l1:
                out["acq_compute_frac_l1ca"] = l1["frac"]
        if acq.get("cpu_baseline"):
            out["acq_cpu_baseline"] = {k: {"ms": v["ms"], "cores": v["cores"], "kind": v["kind"], "gpu_ms": v["gpu_ms"],
                                           "sample": v.get("sample_short", "")}
                                       for k, v in acq["cpu_baseline"].items()}
    spots = result.get("oracle_spot_checks_max_dev_rel_sum_abs_x")
    if spots:
        out["oracle_spot_checks_worst"] = max(spots.values())
    for k in ("device", "compute_units", "libgnsscorr_sha256", "error"):
        if k in result:
            out[k] = result[k]
    if detail_path:
        out["detail"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                      # never reached with today's keys; the contract matters more than the summaries
        for k in ("acq_ms", "configs_if_msps", "ranks_locked", "handover", "rank_diag", "acq_cpu_baseline"):
            out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < LINE_LIMIT and "\n" not in line, len(line)
    return line


def emit(result: dict, detail_path: str | None) -> None:
    """Rank 0: the long form to the detail file (its path on stderr), the short line - the LAST thing on stdout."""
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(result, f)
                f.write("\n")
            print(f"bench.py: full result ({len(json.dumps(result))} bytes) in {detail_path}", file=sys.stderr, flush=True)
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr, flush=True)
            detail_path = None
    sys.stderr.flush()
    try:                                   # libraries that write to C stdio (RCCL's banner) hold their text in a buffer when stdout is a pipe
        import ctypes                      # and would spill it at exit, AFTER the line: flush it out first
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(compact_line(result, detail_path), flush=True)
    try:
        os.dup2(2, 1)                       # whatever a library still writes at exit goes to stderr: the line stays the last on stdout
    except OSError:
        pass


TEARDOWN_LIMIT_S = 30.0


def _finish(result: dict, args, rank: int, closers) -> None:
    """The line first, the tear-down after it.  gc_destroy / destroy_process_group can raise or hang when ranks leave out of step
    (ADVICE r5): a run that finished measuring must still print its line, so rank 0 emits BEFORE anything is closed (emit points
    fd 1 at stderr afterwards: the line stays the last thing on stdout whatever the closers write), every closer's exception is
    reported on stderr instead of raised, and a watchdog ends the process with status 0 if the tear-down does not return."""
    import threading
    if rank == 0:
        emit(result, args.detail)          # the LAST thing this process writes to stdout
    done = threading.Event()

    def _watchdog():
        if not done.wait(TEARDOWN_LIMIT_S):
            print(f"bench.py: tear-down still running after {TEARDOWN_LIMIT_S:.0f} s; the result line is out, leaving", file=sys.stderr, flush=True)
            os._exit(0)
    threading.Thread(target=_watchdog, daemon=True).start()
    for close in closers:
        try:
            close()
        except Exception as e:            # noqa: BLE001 - the measurement is already reported
            print(f"bench.py: {getattr(close, '__qualname__', close)} raised after the result line: {e!r}", file=sys.stderr, flush=True)
    done.set()


def _spawn_ranks(n: int, steps: int, warmup: int) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: one rank per GPU through sharding.launch_ranks (LOCAL_RANK selects
    the device; GC_BENCH_DEVICE pins every rank to one device on a 1-GPU box), rendezvous on 127.0.0.1.  Rank 0 prints the result
    line; every rank's stderr arrives prefixed with its rank.  A rank that dies takes the others with it and the run still ends
    with ONE JSON line (`error`, the failing rank, its stderr tail) - a first 8-GPU run cannot hang without a line."""
    from cu_sdr_collection_amd.sharding import launch_ranks
    out = launch_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], n,
                       rendezvous_timeout_s=float(os.environ.get("GC_RENDEZVOUS_TIMEOUT_S", "240")),
                       total_timeout_s=float(os.environ["GC_BENCH_TIMEOUT_S"]) if "GC_BENCH_TIMEOUT_S" in os.environ else None)
    if out["rc"] != 0:
        r = out["failed_rank"]
        print(_error_line(n, steps, warmup, out["reason"], failed_rank=r, stderr_tail=out["stderr_tail"].get(r, [])[-12:],
                          seconds=round(out["seconds"], 1)), flush=True)
    return out["rc"]


class Ranks:
    """Control plane (barrier, MAX / SUM over ranks: gloo on CPU tensors) and data plane (the record hand-over: RCCL on GPU
    tensors) of the N ranks; a no-op at world size 1.  torch's side of the GPU is initialised here, BEFORE the first Engine:
    torch ships its own copy of the HIP runtime and whichever copy touches the device first serves the process."""

    def __init__(self, rank, world, device=0):
        self.rank, self.world, self.dist, self.device = rank, world, None, device
        self.shared_device = world > 1 and "GC_BENCH_DEVICE" in os.environ     # every rank on ONE GPU: no RCCL communicator possible
        self.rccl = False
        if world > 1 or os.environ.get("GC_BENCH_FORCE_DIST"):
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = "gloo"
            if torch.cuda.is_available():
                torch.cuda.set_device(device)
                torch.cuda.init()
                if not self.shared_device and os.environ.get("GC_BENCH_BACKEND", "rccl") != "gloo":
                    backend, self.rccl = "cpu:gloo,cuda:nccl", True
            import datetime
            # a rank that never arrives (died at gc_create, wrong device) must not hold the others for torch's default 10-30 minutes
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("GC_RENDEZVOUS_TIMEOUT_S", "240"))))
            self.dist = dist
            self.rccl_error = None
            if self.rccl:      # one small all-reduce over RCCL before anything depends on it; every rank learns whether ALL of them got through
                ok = 1.0
                try:
                    x = torch.ones(1, device=f"cuda:{device}")
                    dist.all_reduce(x)
                    torch.cuda.synchronize()
                    ok = float(x.item() == world)
                except Exception as exc:  # noqa: BLE001 - whatever RCCL raises, the bench goes on through the hosts and says so
                    ok, self.rccl_error = 0.0, repr(exc)[:300]
                flag = torch.tensor([ok], dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag[0]) < 1.0:
                    self.rccl = False
                    self.rccl_error = self.rccl_error or "another rank's RCCL probe failed"

    def probe(self):
        """What this rank saw of the node: its device, how many devices its process can see, and how its RCCL probe went.  At world
        size 1 (no process group, torch not imported) the device count is the library's own (gc_device_count)."""
        if self.dist is None:
            from cu_sdr_collection_amd import device_count
            return {"device": self.device, "n_devices_seen": device_count(), "rccl_probe": "not attempted (world size 1)"}
        import torch
        return {"device": self.device, "n_devices_seen": int(torch.cuda.device_count()) if torch.cuda.is_available() else 0,
                "rccl_probe": "ok" if self.rccl else ("not attempted (ranks share one GPU)" if self.shared_device else f"failed: {getattr(self, 'rccl_error', None)}")}

    @property
    def data_backend(self):
        if self.rccl:
            return "nccl (RCCL over xGMI)"
        if self.shared_device:
            return "gloo through the hosts (ranks share one GPU: RCCL needs one GPU per rank)"
        return f"gloo through the hosts (RCCL unavailable: {getattr(self, 'rccl_error', None)})"

    def torch_device(self):
        return f"cuda:{self.device}"

    def barrier(self):
        # an all-reduce of a CPU tensor: routed to gloo by the tensor's device whatever the group's other backend is doing
        # (dist.barrier() on a "cpu:gloo,cuda:nccl" group picks the device itself - the GPU one where it can)
        if self.dist is not None:
            import torch
            self.dist.all_reduce(torch.zeros(1, dtype=torch.float64))

    def reduce(self, values, op="max"):
        if self.dist is None:
            return list(values)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(x) for x in t]

    def gather(self, obj):
        if self.dist is None:
            return [obj]
        # pickled bytes in CPU tensors (gloo), not all_gather_object: that one moves its buffers to the group's "current" device,
        # i.e. through RCCL where a GPU backend exists - the control plane must not depend on the data plane's health
        import pickle
        import torch
        raw = torch.frombuffer(bytearray(pickle.dumps(obj)), dtype=torch.uint8)
        sizes = torch.zeros(self.world, dtype=torch.int64)
        sizes[self.rank] = raw.numel()
        self.dist.all_reduce(sizes)
        width = int(sizes.max())
        mine = torch.zeros(width, dtype=torch.uint8)
        mine[:raw.numel()] = raw
        parts = [torch.zeros(width, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [pickle.loads(bytes(parts[r][:int(sizes[r])].numpy().tobytes())) for r in range(self.world)]

    def turn(self):
        """Context manager around a closed-loop phase: ranks that share one GPU run theirs one after the other (a persistent
        kernel spins until its whole grid is resident, and the library's admission ledger sees its own process only)."""
        import contextlib
        if not self.shared_device:
            return contextlib.nullcontext()
        import fcntl

        @contextlib.contextmanager
        def locked():
            with open(f"/tmp/gc_bench_device{self.device}.lock", "w") as f:
                fcntl.flock(f, fcntl.LOCK_EX)
                try:
                    yield
                finally:
                    fcntl.flock(f, fcntl.LOCK_UN)
        return locked()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def _record_tensor(R: "Ranks", eng, n_samples: int, dtype=np.int8, layout=None):
    """A torch tensor on this rank's GPU holding one IF record, adopted by `eng` (gc_attach_if): what the source rank synthesises
    into and what it then broadcasts."""
    import torch
    from cu_sdr_collection_amd import _lib as L
    comp = 2
    t = torch.empty(n_samples * comp + 64, dtype=torch.int16 if np.dtype(dtype) == np.int16 else torch.int8, device=R.torch_device())
    t[-64:] = 0
    torch.cuda.synchronize()
    eng.attach_if(t.data_ptr(), n_samples, dtype=dtype, layout=L.GC_IQ if layout is None else layout)
    return t


_LIB_SHA = None


def _lib_sha() -> str:
    """SHA-256 of the libgnsscorr.so this process loaded (cu_sdr_collection_amd.build.verify also ties it to the sources)."""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        path = os.path.join(ROOT, "cu-sdr-collection_amd", "lib", "libgnsscorr.so")
        _LIB_SHA = hashlib.sha256(open(path, "rb").read()).hexdigest()
    return _LIB_SHA


def _traffic_entries():
    """Counter passes committed under profiles/ (scripts/collect_profiles.sh + profiles_digest.py: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 corrections of MI355X_MICROARCH.md), newest round first.  An entry counts only when it
    was measured on THIS build (its lib_sha256 = the loaded library's): a kernel change without a re-profile drops the traffic
    figure instead of carrying a stale one."""
    try:
        rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()), key=lambda d: -int(d[1:]))
    except OSError:
        rounds = []
    for rnd in rounds:          # every committed round, newest first (the list used to stop at r03: round 4's passes were never read)
        path = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        try:
            tj = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in (tj if isinstance(tj, list) else [tj]):
            yield rnd, e


def _traffic(kind: str, blocks: int):
    """HBM bytes per launch of a replay kernel from the counter passes of this build: (bytes, source) or (None, why not)."""
    stale = None
    for rnd, e in _traffic_entries():
        if e.get("workload", "l1ca") == kind and e.get("blocks_per_launch") == blocks and e.get("hbm_bytes_per_launch"):
            if e.get("lib_sha256") == _lib_sha():
                return e["hbm_bytes_per_launch"], f"profiles/{rnd}/traffic.json: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of this command on this build (libgnsscorr.so {_lib_sha()[:12]})"
            stale = stale or f"profiles/{rnd}/traffic.json was measured on another build ({str(e.get('lib_sha256'))[:12]}, {e.get('build')}); this one is {_lib_sha()[:12]}: dropped"
    return None, stale


_SHAPE_OF = {"GPS_L5C": "l5", "BDS_B2a": "l5", "GAL_E1C_CBOC": "cboc", "GAL_E1C": "e1x8", "BDS_B1C_NB": "b1c", "GPS_L1CA": "l1ca3"}


def _traffic_scaled(signal: str, channel_samples: float, bytes_per_sample: float):
    """HBM bytes of one replay launch of a BASELINE shape, scaled from the counter passes on the 5-s record of the same shape and
    the same build (FETCH_SIZE x 2 per channel-sample; int16 records scale with the sample size)."""
    shape = _SHAPE_OF.get(signal)
    stale = None
    for rnd, e in _traffic_entries():
        if e.get("workload") == shape and e.get("channel_samples_per_launch") and e.get("hbm_bytes_per_launch"):
            if e.get("lib_sha256") != _lib_sha():
                stale = stale or f"profiles/{rnd}/traffic.json ('{shape}') was measured on another build: dropped"
                continue
            per = e["hbm_bytes_per_launch"] / e["channel_samples_per_launch"] * (bytes_per_sample / 2.0)
            return int(per * channel_samples), f"profiles/{rnd}/traffic.json: {per:.2f} B per channel-sample measured on the 5-s '{shape}' shape of this build, scaled"
    return None, stale


def _cpu_model():
    """Model name and counts of the host CPU the baseline ran on (/proc/cpuinfo)."""
    model, sockets, cores = None, set(), 0
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model is None:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                sockets.add(ln.split(":", 1)[1].strip())
            elif ln.startswith("processor"):
                cores += 1
    except OSError:
        pass
    return {"model": model, "logical_cpus": cores or os.cpu_count(), "sockets": len(sockets) or None}


# =======================================================================================================================
# main workload: GPS L1 C/A x 12
# =======================================================================================================================
def run_l1ca(P, W, args, R: Ranks, device: int):
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd.receiver import track_params
    S = P.initSettings()
    fs, nch = S.samplingFreq, args.channels
    n_samples = int(round(args.seconds * fs))
    n_epochs = int(args.seconds * 1000) - 2
    S.msToProcess, S.numberOfChannels = n_epochs, nch
    eng = P.Engine(device)
    dev_name, cus = eng.device_info()
    # N = 1: the 12 satellites of SURVEY.md §8d config 2.  N > 1: ONE record for all ranks carrying min(32, 12 N) satellites, of
    # which rank r tracks twelve (its channel shard; with more than 32 channels in all, PRNs are tracked by more than one rank)
    n_sats = nch if R.world == 1 else min(32, nch * R.world)
    scene = P.synth.scene(n_sats, 20241008 + 2, fs)
    sats = [scene[(nch * R.rank + i) % n_sats] for i in range(nch)]
    handover, rec_t = None, None
    seed = 20241008 + 2
    if R.dist is not None and not args.no_handover:
        # the sharded path's one exchange step: rank 0 holds the record (synthesised into a torch tensor, standing
        # for fread + upload), broadcasts it over RCCL, and every rank's engine adopts what arrived without a copy
        from cu_sdr_collection_amd.sharding import broadcast_record
        import torch
        t0 = time.time()
        if R.rank == 0:
            rec_t = _record_tensor(R, eng, n_samples)
            P.synth.generate_if_gpu(eng, scene, n_samples, fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=seed, attached=True)
            eng.synchronize()
        t_synth = time.time() - t0
        R.barrier()
        t0 = time.perf_counter()
        rec_t = broadcast_record(rec_t, src=0, device=R.torch_device(), via_host=not R.rccl)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        if R.rank != 0:
            eng.attach_if(rec_t.data_ptr(), n_samples)
        (t_bcast,) = R.reduce([t_bcast])
        nbytes = int(rec_t.numel())
        handover = {"what": "rank 0 -> all ranks: sharding.broadcast_record of the IF record, adopted by gc_attach_if", "backend": R.data_backend,
                    "bytes": nbytes, "seconds": round(t_bcast, 4), "GBps_per_receiver": round(nbytes / t_bcast / 1e9, 2)}
    else:
        t0 = time.time()
        P.synth.generate_if_gpu(eng, scene, n_samples, fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=seed)
        t_synth = time.time() - t0
    eng.set_sampling_freq(fs)
    inits = []
    for i, s in enumerate(sats):   # channel table as preRun would hand it over (truth + a 3 Hz acquisition residual)
        eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
        inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0, code_freq=S.codeFreqBasis,
                                       code_phase=int(np.ceil(s.code_phase_samples)) + 1))
    p = track_params(S)
    with R.turn():
        eng.track(p, inits)                         # first use: code loading, pinned and device allocations (as for the other configs' loops)
        t0 = time.time()
        fields, done, st = eng.track(p, inits)
        t_closed = time.time() - t0
    if st != 0 or int(done.min()) != n_epochs:
        raise RuntimeError(f"closed-loop tracking stopped early: status {st}, epochs {done}")
    locked = np.mean(np.abs(fields["I_P"][:, 1000:]), axis=1) > 3 * np.mean(np.abs(fields["Q_P"][:, 1000:]), axis=1)
    blks = np.ceil((S.codeLength - fields["remCodePhase"]) / (fields["codeFreq"] / fs)).astype(np.int64)
    chan_samples = int(blks.sum())
    with R.turn():
        eng.track(p, inits, device_loop=True)       # first use
        t0 = time.time()
        dfields, ddone, dst = eng.track(p, inits, device_loop=True)
        t_dev = time.time() - t0
    dev_loop = None
    if dst == 0 and int(ddone.min()) == n_epochs:
        diff = dfields["absoluteSample"] != fields["absoluteSample"]
        dev_loop = {"corr_msps": round(chan_samples / t_dev / 1e6, 1), "x_realtime": round(chan_samples / t_dev / 1e6 / nch / (fs / 1e6), 2),
                    "us_per_epoch": round(t_dev / n_epochs * 1e6, 2),
                    # two closed loops part ways by one sample of block boundary at a knife edge of ceil() sooner or later (DESIGN.md
                    # 4.3b; tests/test_gpu_tracking.py::test_four_thousand_epoch_closed_loops_stay_equivalent bounds what follows)
                    "epochs_with_the_host_loops_block_geometry": int(np.argmax(np.any(diff, axis=0))) if diff.any() else int(n_epochs),
                    "max_block_start_difference_samples": int(np.max(np.abs(dfields["absoluteSample"] - fields["absoluteSample"]))),
                    "max_carr_freq_dev_hz": float(np.max(np.abs(dfields["carrFreq"] - fields["carrFreq"])))}
    nb = nch * n_epochs
    job = W.Job("l1ca", W.PACKAGES["GPS_L1CA"], S, sats, eng, params=p, inits=inits, phase0=[0] * nch)
    W.keep_records(job, fields)
    blocks, _ = W.replay_blocks(job)
    eng.replay_prepare(blocks)
    warm = W.warm_engine(P, job) if args.prewarm_ms > 0 else None

    # ---- clocks up (the first ~35 ms of launches after the closed loops' mostly idle device run at lower clocks: 4.4 ms per pass
    #      falling to 3.6), then the W warm-up steps, then exactly K timed steps ------------------------------------------------
    W.clocks_up(warm, args.prewarm_ms)
    for _ in range(args.warmup):
        eng.replay_launch()
    eng.synchronize()
    R.barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        eng.replay_launch()
    kernel_ms_total = eng.timer_stop()  # hipEvents on the launch stream; also synchronises
    eng.synchronize()
    R.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, kernel_ms_total = R.reduce([elapsed, kernel_ms_total])
    (total_chan_samples,) = R.reduce([chan_samples], op="sum")

    out = eng.replay_fetch()[:, 0, :]
    rec = np.stack([fields[f].T.reshape(-1) for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")], axis=1)
    scale = 2.0 * 18000 * 28.0
    replay_dev = float(np.max(np.abs(out - rec)) / scale)

    ms_per_step = elapsed * 1e3 / args.steps
    corr_msps = total_chan_samples * args.steps / elapsed / 1e6
    value = corr_msps / nch                     # IF samples per second with all channels active, all ranks
    kernel_ms = kernel_ms_total / args.steps
    algo_bytes = 2.0 * chan_samples             # int8 I/Q: 2 bytes per channel-sample
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = _traffic("l1ca", nb)
    result = {
        "metric": METRIC, "value": round(value, 1), "unit": "IF Msamples/s (all channels active, all GPUs)",
        "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 accumulate over int8 I/Q samples; f64/64-bit fixed-point code+carrier phase", "data": "synthetic",
        "config": {"workload": f"GPS L1 C/A, {nch} channels/GPU, 1-ms E/P/L correlators, {args.seconds:g} s of int8 I/Q IF at 18 Msps "
                               f"({n_samples * 2 / 1e9:.2f} GB in HBM), batched replay of {nb} blocks per step",
                   "channels_per_gpu": nch, "epochs": n_epochs, "blocks_per_step": nb, "prewarm_ms": args.prewarm_ms,
                   "parallelism": (f"channels sharded over {R.world} GPU(s); one exchange step: the record hand-over ({R.data_backend})" if handover
                                   else f"channels sharded over {R.world} GPU(s), every rank synthesises its own copy of the record (no hand-over)")},
        "corr_msps": round(corr_msps, 1), "corr_msps_unit": "Msamples/s (channel-samples, all GPUs)",
        "if_msps_per_gpu": round(value / R.world, 1), "x_realtime_replay": round(value / R.world / (fs / 1e6), 1),
        "closed_loop": {"corr_msps": round(chan_samples / t_closed / 1e6, 1), "x_realtime": round(chan_samples / t_closed / 1e6 / nch / (fs / 1e6), 2),
                        "us_per_epoch": round(t_closed / n_epochs * 1e6, 2), "channels_locked": int(locked.sum())},
        "closed_loop_device": dev_loop,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "corr_epl_fast_kernel<ARMS=1, I8_IQ, SPL=16> (four waves, float tables)", "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": algo_bytes},
        "replay_vs_closed_loop_max_dev": replay_dev, "device": dev_name, "compute_units": cus, "synth_s": round(t_synth, 2),
        "libgnsscorr_sha256": _lib_sha(),
    }
    if handover:
        result["handover"] = handover
    if R.world > 1:
        lock = R.gather({"rank": R.rank, "channels_locked": int(locked.sum()), "prns": [s.prn for s in sats], "replay_vs_closed_loop_max_dev": replay_dev,
                         "kernel_ms": round(kernel_ms, 4), "frac": round(achieved / HBM_PEAK_GBPS, 4), **R.probe()})
        result["ranks"] = lock
        result["rank_diag"] = [{k: r.get(k) for k in ("rank", "device", "n_devices_seen", "rccl_probe")} for r in lock]
    else:       # the same self-diagnosis at N = 1: what the one rank saw of the node (the driver's first 8-GPU run reads the same keys per rank)
        result["rank_diag"] = [{"rank": 0, **R.probe()}]
    return result, dict(eng=eng, S=S, sats=sats, scene=scene, inits=inits, fields=fields, n_epochs=n_epochs, n_samples=n_samples, job=job, record=rec_t)


ACQ_WARMUP_CALLS = 8   # untimed searches in front of the timed ones (run_acquisition, run_acquisition_packages)


def run_acquisition(P, eng, sats, R: Ranks):
    """The other half of the hot path at the reference's defaults: 32 PRNs x 29 bins x 20 ms (acquisition.m:116-260).  N > 1: the
    PRN list shards over the ranks (sharding.shard_prns: every rank transforms the record's 20 hops itself - they are hoisted out
    of the PRN loop - and searches its share); rank 0 merges the acqResults (merge_acq_results, disjoint shares: exact)."""
    from cu_sdr_collection_amd.receiver import acquisition as gpu_acquisition
    from cu_sdr_collection_amd.sharding import merge_acq_results, shard_prns
    Sa = P.initSettings()
    full = list(Sa.acqSatelliteList)
    Sa.acqSatelliteList = shard_prns(full, R.world, R.rank)
    gpu_acquisition(eng, Sa)  # first use (plans, twiddles, scratch)
    eng.synchronize()
    R.barrier()
    # untimed warm-up calls: a 3-ms call straight after an idle phase runs ~10 % slower than the same call in a sequence of searches
    # (scripts/acq_clock_probe.py: 3.7, 3.6, 3.3, 3.2 ... 3.06 ms over the first ~10 calls; the sensors show no clock or power limit,
    # 290 W of 1 400) - a receiver acquires band after band, so the sequence is the figure
    for _ in range(ACQ_WARMUP_CALLS):
        gpu_acquisition(eng, Sa)
    calls = []
    for _ in range(5):            # five timed calls, the median reported
        t0 = time.perf_counter()
        eng.timer_start()
        acq = gpu_acquisition(eng, Sa)
        ev = eng.timer_stop()     # hipEvents on the engine's stream around the whole call (its kernels + the gaps of its three host steps)
        calls.append((time.perf_counter() - t0, ev))
    calls.sort()
    t_acq, ev_ms = calls[len(calls) // 2]
    (t_acq,) = R.reduce([t_acq])
    acq = merge_acq_results(R.gather(acq))
    found = sorted(int(i) + 1 for i in np.nonzero(acq.carrFreq)[0])
    truth = {s.prn: s for s in sats}
    phase_ok = all(abs(((int(acq.codePhase[p - 1]) - 1 - truth[p].code_phase_samples + 9000) % 18000) - 9000) <= 2.0 for p in found if p in truth)
    # Roofline of the search (acquisition.m:167-200 is the dominant row).  Algorithmic HBM bytes: what the search has
    # to read and write at all - the (H + 1) code periods of int8 I/Q signal, one sampled code per PRN, the fine stage's 40 code
    # periods per detection, a few words of result per PRN.  Algorithmic flops: one forward transform per hop, one per code, one
    # inverse transform + product + |.| + accumulation per (PRN, bin, hop), 5 N log2 N per transform.  Everything between - the
    # hop spectra, the rows pass's output the columns pass reads back (334 MB per PRN) - is traffic of THIS
    # implementation, not of the problem: the byte figure shows how far from its input the search is, the flop figure how
    # far from the vector units (157.3 TFLOP/s f32, MI355X_MICROARCH.md; FFT butterflies do not map to MFMA tiles).
    import math
    n_fft, hops, bins, nprn = 36000, 20, 29, len(Sa.acqSatelliteList)
    spc = n_fft // 2
    fft_flops = 5.0 * n_fft * math.log2(n_fft)
    flops = (hops + nprn) * fft_flops + nprn * bins * hops * (fft_flops + 6.0 * n_fft + 4.0 * n_fft)
    ndet = int(np.count_nonzero(acq.carrFreq))
    abytes = (hops + 1) * spc * 2.0 + nprn * spc * 1.0 + ndet * 40 * spc * 2.0 + nprn * 32.0
    roof = {"bound": "hbm", "achieved": round(abytes / (ev_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(abytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6), "traffic": None, "kernel_ms": round(ev_ms, 4),
            "algorithmic_bytes": abytes, "kernels": "fft_pass_ct<...> (forward, code, inverse rows / columns + |.| + hop sums), abs_combine_kernel, fine_multi_kernel",
            "compute": {"bound": "valu f32", "achieved": round(flops / (ev_ms * 1e-3) / 1e12, 3), "peak": 157.3, "unit": "TFLOP/s",
                        "frac": round(flops / (ev_ms * 1e-3) / 1e12 / 157.3, 4), "algorithmic_flops": flops},
            "note": "rank 0's share of the PRN list; the search is bound by neither figure yet: its passes bounce the inverse transforms' intermediate through "
                    "the memory system (DESIGN.md §4.4)"}
    return {"seconds": round(t_acq, 5), "seconds_of_the_timed_calls": [round(w, 5) for w, _ in calls], "warmup_calls": 1 + ACQ_WARMUP_CALLS, "roofline": roof, "prns_searched": len(full), "prns_per_rank": len(Sa.acqSatelliteList), "bins": 29, "non_coh_ms": 20, "fft_size": 36000,
            "acquired": found, "all_scene_prns_found": sorted(truth) == found, "code_phases_within_two_samples_of_the_scene": bool(phase_ok),
            "sharding": f"PRN list round-robin over {R.world} rank(s)"}


def run_acquisition_packages(P, device, only=None):
    """acquisition.m of all twelve packages at the reference's DEFAULT search sizes (settings = initSettings() unmodified), each on
    the record of its committed fixture tests/golden/ref_acq_<pkg>_default.npz (the reference's own acquisition.m executed on that
    record by oracle/mlab, minutes per package): codePhase / carrFreq must be array_equal to the fixture - the search that is timed
    is the search that is checked.  Per package: wall and hipEvent milliseconds of one whole call (median of three) (searches + fine stage + the host
    steps between them), the transforms it stands for (signal spectra hoisted: forward = bins x hops, one per code, one inverse
    per PRN x arm x bin x hop), the transform length, and the algorithmic flops (5 N log2 N per transform + 10 N per inverse for
    product, |.| and accumulation) against the f32 vector peak."""
    import math
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_scenes as RS
    out = {}
    for sc in RS.DEFAULT_ACQ_SCENES:
        name = sc.name[:-len("_default")]
        if only and name not in only:
            continue
        path = os.path.join(ROOT, "tests", "golden", f"ref_acq_{sc.name}.npz")
        if not os.path.exists(path):
            out[name] = {"error": "no fixture"}
            continue
        z = np.load(path)
        S, rec = RS.acq_inputs(P, sc)
        if RS.crc(rec) != int(z["record_crc32"][0]):
            out[name] = {"error": "the rebuilt record is not the fixture's"}
            continue
        with P.Engine(device) as eng:
            eng.load_if(rec, fs=S.samplingFreq)
            sc.product(P, eng, S)                      # first use: plans, twiddles, scratch, code tables
            eng.synchronize()
            t0, spent, warm = time.perf_counter(), 0.0, 0
            while warm < ACQ_WARMUP_CALLS and spent < 0.03:   # untimed calls until ~30 ms of searches have run (see run_acquisition)
                sc.product(P, eng, S)
                warm += 1
                spent = time.perf_counter() - t0
            runs = []
            for _ in range(3):                         # three timed calls, the median reported
                eng.acq_stats.clear()
                t0 = time.perf_counter()
                eng.timer_start()
                got = sc.product(P, eng, S)
                ev = eng.timer_stop()
                runs.append((time.perf_counter() - t0, ev))
            runs.sort()
            wall, ev_ms = runs[1]
            st = dict(eng.acq_stats)
            try:
                guard = eng.acq_guard_stats()       # the float64 guard of the LAST search call of the package's acquisition (csrc/acq_guard.h)
            except Exception:                      # noqa: BLE001
                guard = None
        same = {f: bool(np.array_equal(np.asarray(getattr(got, f), dtype=np.float64), z["f_" + f])) for f in sc.fields if f != "peakMetric"}
        want = z["f_peakMetric"]
        metric_dev = float(np.max(np.abs(np.asarray(got.peakMetric, dtype=np.float64) - want)) / np.max(np.abs(want)))
        n = st.get("n_fft", 0)
        nt = st.get("forward", 0) + st.get("code", 0) + st.get("inverse", 0)
        flops = nt * 5.0 * n * math.log2(max(n, 2)) + st.get("inverse", 0) * 10.0 * n
        out[name] = {"ms": round(wall * 1e3, 3), "event_ms": round(ev_ms, 3), "ms_of_three_calls": [round(w * 1e3, 3) for w, _ in runs], "warmup_calls": 1 + warm, "fft_size": n, "transforms": nt, "inverse_transforms": st.get("inverse", 0),
                     "prns": len(list(S.acqSatelliteList)), "detected": int(np.count_nonzero(z["f_carrFreq"])),
                     "equal_to_the_references_acquisition_m": same, "peak_metric_max_rel_dev": round(metric_dev, 7),
                     "ms_per_prn": round(wall * 1e3 / max(1, len(list(S.acqSatelliteList))), 4),
                     "reference_interpreter_seconds": round(float(z["seconds"][0]), 1) if "seconds" in z.files else None,
                     "float64_guard": guard,
                     "roofline": {"compute": {"bound": "valu f32", "achieved": round(flops / (ev_ms * 1e-3) / 1e12, 3), "peak": 157.3, "unit": "TFLOP/s",
                                              "frac": round(flops / (ev_ms * 1e-3) / 1e12 / 157.3, 4), "algorithmic_flops": flops}}}
    return out


def run_closed_loop_sweep(P, W, device, seconds: float, l1_points=(12, 24, 48, 96, 192), l5_points=(16, 32, 64)):
    """north_star: "shard across the 8 GPUs ... only when the channel count warrants it" (tracking.m:133: the reference's channel
    loop).  Closed loops - the mode a receiver runs in - of n channels on ONE GPU over one record: seconds per call, microseconds
    per epoch, x real time, which launcher the library chose (gc_debug_last_track_mode: 1 persistent host-fed kernel, 0 a launch
    per epoch, 2 loop closed on the device), and the share of the wall time the GPU needs for the same blocks in one batched
    replay launch (`gpu_busy_estimate`: what is left is latency - PCIe round trips, the all-gather of the partial sums).  The
    channels beyond the 32 satellites of the scene track satellites other channels track too: the same work per channel.
    `knee_channels`: the largest measured count whose epoch time is within 1.25x of the smallest count's - below it more GPUs buy
    nothing (every channel's epoch is its own latency chain), above it the epoch time grows with the channels and sharding
    divides it: sharding.recommended_world_size is built on these knees."""
    import copy
    out = {}
    for name, pkgname, fs, points in (("GPS_L1CA_18Msps", "GPS_L1CA", 18e6, l1_points), ("GPS_L5C_50Msps", "GPS_L5C", 50e6, l5_points)):
        owner = P.Engine(device)
        (pkg, S, scene), = W.make_band(P, owner, [(pkgname, 32)], seconds, fs, 20e3, 7007)
        n_ep = int((seconds - 3 * S.intTime) / S.intTime) - 1
        rows = []
        for n in points:
            eng = P.Engine(device)
            try:
                eng.share_if(owner)
                eng.set_sampling_freq(fs)
                sats = [scene[i % len(scene)] for i in range(n)]
                job = W.prepare_job(P, W.Job(f"{name} x {n}", pkg, copy.copy(S), sats, eng), n_ep)
                row = {"channels": n}
                recs = None
                for key, device_loop in (("host_closed", False), ("device_closed", True)):
                    try:
                        W.run_closed_loops(P, [job], device_loop=device_loop)                  # first use
                        t, recs = W.run_closed_loops(P, [job], device_loop=device_loop)
                        row[key] = {"seconds": round(t, 4), "us_per_epoch": round(t / n_ep * 1e6, 2), "x_realtime": round(n_ep * S.intTime / t, 1),
                                    "launcher": {0: "launch per epoch", 1: "persistent host-fed kernel", 2: "device loop"}.get(eng.last_track_mode(), "?")}
                    except Exception as exc:  # noqa: BLE001 - a count the library cannot take is a finding of the sweep, not an error of the bench
                        row[key] = {"error": str(exc)[:160]}
                if recs is not None:
                    W.keep_records(job, recs[0])
                    row["locked"] = W.locked(job)
                    ms, _, kern = W.time_replay(job, 3, 1, prewarm_ms=0.0)
                    row["replay_ms"] = round(ms, 4)
                    row["replay_kernel"] = W.KERNEL_NAMES.get(kern, str(kern))
                    for key in ("host_closed", "device_closed"):
                        if "seconds" in row[key]:
                            row[key]["gpu_busy_estimate"] = round(ms * 1e-3 / row[key]["seconds"], 4)
                rows.append(row)
            finally:
                eng.close()
        owner.close()
        knee = {}
        for key in ("host_closed", "device_closed"):
            ok = [r for r in rows if "us_per_epoch" in r[key]]
            if ok:
                base = ok[0][key]["us_per_epoch"]
                knee[key] = max(r["channels"] for r in ok if r[key]["us_per_epoch"] <= 1.25 * base)
        out[name] = {"record_seconds": seconds, "epochs_per_channel": n_ep, "points": rows, "knee_channels": knee}
    from cu_sdr_collection_amd import sharding
    out["policy"] = {"what": "sharding.recommended_world_size(n_channels, signal): 1 up to the knee, then ceil(n / knee) GPUs (at most 8)",
                     "table_in_sharding_py": sharding.CLOSED_LOOP_KNEE,
                     "examples": {f"{sig} x {n}": sharding.recommended_world_size(n, sig) for sig in ("GPS_L1CA", "GPS_L5C") for n in (12, 64, 192, 512)}}
    return out


# =======================================================================================================================
# the other BASELINE configs: a band record, one job per package, closed loops + replay
# =======================================================================================================================
def run_band_jobs(P, W, name, device, parts, seconds, fs, intermediate_freq, seed, steps, warmup, dtype=np.int8, side_by_side=True):
    """parts = [(package, channels)] tracked from ONE record.  Returns (summary dict, jobs)."""
    engines = [P.Engine(device) for _ in parts]
    t0 = time.time()
    made = W.make_band(P, engines[0], parts, seconds, fs, intermediate_freq, seed, dtype=dtype)
    t_synth = time.time() - t0
    for e in engines[1:]:
        e.share_if(engines[0])
    jobs = []
    for (pkg, S, sats), eng in zip(made, engines):
        n_ep = int((seconds - 3 * S.intTime) / S.intTime) - 1
        j = W.prepare_job(P, W.Job(f"{name}:{pkg.signal}", pkg, S, sats, eng), n_ep)
        j.record_dtype = np.dtype(dtype)
        jobs.append(j)
    W.run_closed_loops(P, jobs, device_loop=False)                      # first use: code loading, allocations
    t_host, recs = W.run_closed_loops(P, jobs, device_loop=False)
    for j, f in zip(jobs, recs):
        W.keep_records(j, f)
    W.run_closed_loops(P, jobs, device_loop=True)                       # (falls back to the host loop where no device-loop kernel applies)
    t_dev, _ = W.run_closed_loops(P, jobs, device_loop=True)
    per_job = []
    total_cs, total_ms, total_bytes = 0.0, 0.0, 0.0
    total_traffic = 0
    bps = 2.0 * np.dtype(dtype).itemsize
    for j in jobs:
        warm = W.warm_engine(P, j)
        ms, dev, kern = W.time_replay(j, steps, warmup, warm=warm)
        warm.close()
        cs = float(j.blks.sum())
        total_cs += cs
        total_ms += ms
        total_bytes += bps * cs
        tr, tr_src = _traffic_scaled(j.pkg.signal, cs, bps)
        total_traffic = None if (tr is None or total_traffic is None) else total_traffic + tr
        per_job.append({"signal": j.pkg.signal, "channels": len(j.sats), "epochs": j.params.n_epochs, "blocks_per_launch": int(j.blks.size),
                        "traffic": tr, "traffic_source": tr_src,
                        "kernel": W.KERNEL_NAMES.get(kern, str(kern)), "kernel_ms": round(ms, 4),
                        "algorithmic_bytes_per_launch": bps * cs, "achieved_GBps": round(bps * cs / ms / 1e6, 1),
                        "frac": round(bps * cs / ms / 1e6 / HBM_PEAK_GBPS, 4), "replay_vs_closed_loop_max_dev": dev,
                        "channels_locked": W.locked(j)})
    nch = sum(len(j.sats) for j in jobs)
    signal_s = min(j.params.n_epochs * j.S.intTime for j in jobs)
    together = None
    if len(jobs) > 1 and side_by_side:   # a band's packages replayed side by side (their own streams), as gc_track_multi runs their closed loops
        ms_all = W.time_replays_together(jobs, max(steps, 10), warmup)
        together = {"ms_per_pass": round(ms_all, 4), "frac": round(total_bytes / ms_all / 1e6 / HBM_PEAK_GBPS, 4), "clock": "host, around all streams",
                    "note": "all jobs of the band launched per pass, each on its context's stream; kernel_ms_sum above is the jobs one after the other"}
    summary = {
        "workload": f"{name}: {', '.join(f'{n} x {s}' for s, n in parts)} on one {seconds:g}-s {np.dtype(dtype).name} I/Q record at {fs / 1e6:g} Msps "
                    f"({int(round(seconds * fs)) * bps / 1e9:.2f} GB in HBM)",
        "channels": nch, "jobs": per_job,
        "replay": {"kernel_ms_sum": round(total_ms, 4), "corr_msps": round(total_cs / total_ms / 1e3, 1), "if_msps": round(total_cs / total_ms / 1e3 / nch, 1),
                   "x_realtime": round(total_cs / total_ms / 1e3 / nch / (fs / 1e6), 1),
                   "roofline": {"bound": "hbm", "achieved": round(total_bytes / total_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(total_bytes / total_ms / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": total_bytes,
                                "traffic": total_traffic or None},
                   "side_by_side": together},
        "closed_loop_host": {"seconds": round(t_host, 4), "x_realtime": round(signal_s / t_host, 1)},
        "closed_loop_device": {"seconds": round(t_dev, 4), "x_realtime": round(signal_s / t_dev, 1)},
        "synth_s": round(t_synth, 2),
    }
    return summary, jobs


def run_mix(P, W, args, R: Ranks, device: int):
    """BASELINE configs[4]: 64 channels of the twelve signals sharded over the ranks by band (sharding.shard_bands: every rank
    8 channels, each band's record on as few GPUs as possible).  The hand-over of a real run, end to end: for every band the first
    rank that tracks one of its channels holds the record (synthesised into a torch tensor on its GPU, standing for fread +
    upload), sharding.distribute_band_records broadcasts it to the band's other ranks - one process group per band, RCCL over
    xGMI - and every rank's engines adopt what arrived (gc_attach_if).  Then one tracking job per package, all jobs of a rank
    concurrently (gc_track_multi).  --no-handover: every rank of a band synthesises its own copy from the same seed."""
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd.sharding import band_ranks, distribute_band_records, shard_bands
    counts = {b: sum(n for _, n in parts) for b, parts in W.MIX_BANDS.items()}
    plan = shard_bands(counts, R.world)
    mine = plan[R.rank]
    readers = band_ranks(plan)
    band_fs = {"L2": 8e6, "GLO_L1": 12e6, "GLO_L2": 12e6}
    band_if = {"GLO_L1": 0.0, "GLO_L2": 0.0}
    band_order = sorted(W.MIX_BANDS)
    fs_of = lambda b: band_fs.get(b, 18e6)
    seed_of = lambda b: 5000 + 101 * band_order.index(b)
    nsamp_of = lambda b: int(round(args.mix_seconds * fs_of(b)))
    first_engine, t_synth = {}, [0.0]
    handover = None
    if R.dist is not None and not args.no_handover:
        import torch

        def read_record(band):           # runs on the band's first rank only
            eng = P.Engine(device)
            t0 = time.time()
            t = _record_tensor(R, eng, nsamp_of(band), layout=L.GC_QI if W.band_is_qi(W.MIX_BANDS[band]) else L.GC_IQ)
            W.make_band(P, eng, W.MIX_BANDS[band], args.mix_seconds, fs_of(band), band_if.get(band, 20e3), seed_of(band), attached=True)
            eng.synchronize()
            t_synth[0] += time.time() - t0
            first_engine[band] = eng
            return t

        timings = {}
        R.barrier()
        t0 = time.perf_counter()
        records = distribute_band_records(plan, read_record, device=R.torch_device(), via_host=not R.rccl, timings=timings)
        torch.cuda.synchronize()
        R.barrier()
        t_all = time.perf_counter() - t0 - t_synth[0]
        per_band = {b: {"seconds": round(dt, 4), "bytes": nb, "GBps": round(nb / dt / 1e9, 2)} for b, (dt, nb) in timings.items()}
        handover = {"what": "per band: first rank -> the band's other ranks (sharding.distribute_band_records, one process group per band), adopted by gc_attach_if",
                    "backend": R.data_backend, "bands_received_or_sent_here": per_band, "seconds_all_bands_this_rank": round(max(t_all, 0.0), 4)}
    jobs, bands_here = [], []
    for band in band_order:
        idx = sorted(i for b, i in mine if b == band)
        if not idx:
            continue
        bands_here.append(band)
        parts = W.MIX_BANDS[band]
        fs = fs_of(band)
        if handover is not None:
            made, _ = W.band_scene(P, parts, fs, seed_of(band))
            eng0 = first_engine.get(band)
            if eng0 is None:               # the record arrived over the wire
                eng0 = P.Engine(device)
                eng0.attach_if(records[band].data_ptr(), nsamp_of(band), layout=L.GC_QI if W.band_is_qi(parts) else L.GC_IQ)
            eng0.set_sampling_freq(fs)
            eng0._record_tensor = records[band]          # keep the tensor alive as long as the engine reads it
            engines = [eng0] + [P.Engine(device) for _ in parts[1:]]
        else:
            engines = [P.Engine(device) for _ in parts]
            t0 = time.time()
            made = W.make_band(P, engines[0], parts, args.mix_seconds, fs, band_if.get(band, 20e3), seed_of(band))
            t_synth[0] += time.time() - t0
        off = 0
        owner_used = False
        for (pkg, S, sats), eng in zip(made, engines):
            sel = [s for k, s in enumerate(sats) if off + k in idx]       # this rank's channels of the package
            off += len(sats)
            if not sel:
                if eng is not engines[0]:
                    eng.close()
                continue
            if eng is not engines[0]:
                eng.share_if(engines[0])
            n_ep = int((args.mix_seconds - 3 * S.intTime) / S.intTime) - 1
            jobs.append(W.prepare_job(P, W.Job(f"{band}:{pkg.signal}", pkg, S, sel, eng), n_ep))
            jobs[-1].record_dtype = np.dtype(np.int8)
            jobs[-1].swap_iq = W.band_is_qi(parts)
    with R.turn():
        W.run_closed_loops(P, jobs, device_loop=False)
        t_host, recs = W.run_closed_loops(P, jobs, device_loop=False)
    for j, f in zip(jobs, recs):
        W.keep_records(j, f)
    with R.turn():
        W.run_closed_loops(P, jobs, device_loop=True)
        t_dev, drecs = W.run_closed_loops(P, jobs, device_loop=True)
    per_job, cs_total, bytes_total = [], 0.0, 0.0
    for j in jobs:
        blocks, _ = W.replay_blocks(j)
        j.engine.replay_prepare(blocks)
        cs = float(j.blks.sum())
        cs_total += cs
        bytes_total += 2.0 * cs
    for _ in range(args.warmup):
        for j in jobs:
            j.engine.replay_launch()
    for j in jobs:
        j.engine.synchronize()
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):          # a step = one replay pass of every job of the rank; the jobs' streams overlap
        for j in jobs:
            j.engine.replay_launch()
    for j in jobs:
        j.engine.synchronize()
    R.barrier()
    elapsed = time.perf_counter() - t0
    (elapsed,) = R.reduce([elapsed])
    cs_all, nch_all, bytes_all = R.reduce([cs_total, float(sum(len(j.sats) for j in jobs)), bytes_total], op="sum")
    kernel_ms = 0.0
    spots = {}
    if args.spot_check:                   # CPU side, after every timed region: the only place of this function that touches oracle/
        from oracle import gnss_oracle as O
        for j in jobs:
            spots[j.name] = oracle_spot_check(P, W, O, j, nblocks=3)
    for j, df in zip(jobs, drecs):        # per-kernel times, one job at a time (hipEvents on the job's stream)
        j.engine.timer_start()
        for _ in range(args.steps):
            j.engine.replay_launch()
        ms = j.engine.timer_stop() / args.steps
        kernel_ms += ms
        cs = float(j.blks.sum())
        out = j.engine.replay_fetch()[:, 0, :]
        rec = np.stack([j.fields[f].T.reshape(-1) for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")], axis=1)
        replay_dev = float(np.max(np.abs(out - rec)) / (2.0 * float(j.blks.mean()) * 28.0))
        dj = W.Job(j.name, j.pkg, j.S, j.sats, j.engine, params=j.params)
        W.keep_records(dj, df)
        per_job.append({"job": j.name, "channels": len(j.sats), "epochs": j.params.n_epochs, "kernel": W.KERNEL_NAMES.get(j.engine.last_kernel()),
                        "kernel_ms": round(ms, 4), "achieved_GBps": round(2.0 * cs / ms / 1e6, 1), "frac": round(2.0 * cs / ms / 1e6 / HBM_PEAK_GBPS, 4),
                        "channels_locked": W.locked(j), "channels_locked_device_loop": W.locked(dj), "replay_vs_closed_loop_max_dev": replay_dev,
                        "oracle_spot_check_max_dev_rel_sum_abs_x": spots.get(j.name)})
    name, cus = jobs[0].engine.device_info()
    corr_msps = cs_all * args.steps / elapsed / 1e6
    achieved = bytes_total / (kernel_ms * 1e-3) / 1e9
    mine_summary = {"rank": R.rank, "bands": bands_here, "jobs": per_job, "closed_loop_host_s": round(t_host, 4), "closed_loop_device_s": round(t_dev, 4),
                    "handover": handover}
    gathered = R.gather(mine_summary)
    t_host_max, t_dev_max = R.reduce([t_host, t_dev])
    all_jobs = [j for r in gathered for j in r["jobs"]]
    result = {
        "metric": METRIC, "value": round(corr_msps / (nch_all / R.world), 1), "unit": "IF Msamples/s (channel-samples / channels per GPU, all GPUs)",
        "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32 accumulate over int8 I/Q samples; f64/64-bit fixed-point code+carrier phase",
        "data": "synthetic",
        "config": {"workload": f"all-constellation mix: {int(nch_all)} channels of {len(W.PACKAGES) - 1} signals in {len(W.MIX_BANDS)} band records "
                               f"({args.mix_seconds:g} s each), sharded by band over {R.world} GPU(s), every rank's packages tracked concurrently",
                   "bands": {b: parts for b, parts in W.MIX_BANDS.items()}, "plan_bands_per_rank": {str(r): sorted({b for b, _ in p}) for r, p in enumerate(plan)},
                   "band_reader_ranks": readers,
                   "parallelism": (f"channels sharded by IF record; exchange step: the band records' hand-over ({R.data_backend})" if handover
                                   else "channels sharded by IF record; every rank of a band synthesises its own copy (no hand-over)")},
        "corr_msps": round(corr_msps, 1),
        "channels": int(nch_all), "channels_locked": int(sum(j["channels_locked"] for j in all_jobs)),
        "channels_locked_device_loop": int(sum(j["channels_locked_device_loop"] for j in all_jobs)),
        "signals": sorted({j["job"].split(":")[1] for j in all_jobs}),
        "closed_loop_host": {"seconds": round(t_host_max, 4), "x_realtime": round(args.mix_seconds / t_host_max, 1)},
        "closed_loop_device": {"seconds": round(t_dev_max, 4), "x_realtime": round(args.mix_seconds / t_dev_max, 1)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": None, "kernel": "sum over rank 0's replay kernels (per job under ranks)", "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": bytes_total},
        "ranks": gathered, "device": name, "compute_units": cus, "synth_s": round(t_synth[0], 2),
    }
    return result


# =======================================================================================================================
# CPU side (rank 0, N = 1 only, after every timed region): baselines and oracle spot checks.  The only place that touches oracle/.
# =======================================================================================================================
def _oracle_tables(O, P, signal, prn, code_length=10230):
    pad = O.pad_code
    if signal == "GAL_E1C_CBOC":   # the BOC(6,1) table of the E1-C pilot is this build's extension (no oracle generator in the reference's terms)
        return [pad(O.generate_e1_code(prn, "B")), pad(O.generate_e1_code(prn, "C")), pad(np.asarray(P.codes.generateE1C_BOC61(prn), dtype=np.float64))]

    def e5(i_sig, q_sig):          # GAL_E5a tracking.m:148-156: the data table is the head of the TIERED code, the pilot the primary code
        tiered = O.generate_e5_code(i_sig, prn, 2)
        return [np.concatenate([[tiered[code_length - 1]], tiered, [tiered[0]]])[:code_length + 2], pad(O.generate_e5_primary(q_sig, prn))]

    return {
        "GPS_L1CA": lambda: [pad(O.generate_ca_code(prn))],
        "GAL_E1C": lambda: [pad(O.generate_e1_code(prn, "B")), pad(O.generate_e1_code(prn, "C"))],
        "BDS_B1C_NB": lambda: [pad(O.generate_b1c_code(prn, "data")), pad(O.generate_b1c_code(prn, "pilot11"))],
        "GPS_L5C": lambda: [pad(O.generate_l5_code(prn, "I")), pad(O.generate_l5_code(prn, "Q"))],
        "BDS_B2a": lambda: [pad(O.generate_b2a_code(prn, "data")), pad(O.generate_b2a_code(prn, "pilot"))],
        "BDS_B1I": lambda: [pad(O.generate_b1i_code(prn))],
        "BDS_B3I": lambda: [pad(O.generate_b3i_code(prn))],
        "GAL_E5a": lambda: e5("e5ai", "e5aq"),
        "GAL_E5b": lambda: e5("e5bi", "e5bq"),
        "GLO_GL1": lambda: [pad(O.generate_glo_code())],
        "GLO_GL2": lambda: [pad(O.generate_glo_code())],
        "GPS_L2C": lambda: [pad(O.generate_l2c_code(prn, "CM", 10230))],   # the CM arm (the CL arm's 1.5-s table is not regenerated here)
    }[signal]()


def oracle_spot_check(P, W, O, job, nblocks=4, seed=1):
    """A few replayed blocks of a job against the float64 oracle (tracking.m:247-300) at identical descriptors: worst deviation in
    units of sum |x| over the block (the tests' tolerance is 2e-6)."""
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd import signals
    spec = signals.SIGNALS[job.pkg.signal]
    rng = np.random.default_rng(seed)
    _, view = W.replay_blocks(job)
    pick = np.sort(rng.choice(view.shape[0], size=nblocks, replace=False))
    sub = job.engine.make_blocks(nblocks)
    np.frombuffer(sub, dtype=W.BLOCK_DT)[:] = view[pick]
    got = job.engine.correlate(sub)
    swap = bool(getattr(job, "swap_iq", False))         # GLONASS records hold Q,I (GLO_GL1 tracking.m:227)
    worst = 0.0
    for k in range(nblocks):
        b = sub[k]
        tabs = _oracle_tables(O, P, job.pkg.signal, job.sats[b.channel].prn)
        raw_i = job.engine.read_if(int(b.first_sample), int(b.blksize), dtype=getattr(job, "record_dtype", np.int8), layout=L.GC_QI if swap else L.GC_IQ)
        raw = O.raw_from_if(raw_i, 0, int(b.blksize), swap_iq=swap)
        want, _, _ = O.correlate_block(raw, tabs, b.rem_code_phase, b.code_phase_step, b.el_spacing, b.carr_freq, b.rem_carr_phase,
                                       job.params.sampling_freq, job.params.code_length, r=spec.index_scale,
                                       arm_mult=list(spec.arm_mult)[:len(tabs)] if spec.arm_mult else None)
        scale = float(np.sum(np.abs(raw.real)) + np.sum(np.abs(raw.imag)))
        worst = max(worst, float(np.max(np.abs(got[k, :want.shape[0]] - want))) / scale)
    return worst


def cpu_leg(P, W, args, main, extra_jobs):
    from types import SimpleNamespace

    from oracle import c_oracle as CO
    from oracle import gnss_oracle as O
    eng, fields, inits, n_epochs, n_samples = main["eng"], main["fields"], main["inits"], main["n_epochs"], main["n_samples"]
    nch = len(inits)
    fs = 18e6
    CO.build(force=True)
    cpu_epochs = min(args.cpu_epochs, n_epochs)
    iq = eng.read_if(0, min(int((cpu_epochs + 3) * 1e-3 * fs), n_samples))
    Sc = P.initSettings()
    Sc.msToProcess = cpu_epochs
    ch = [SimpleNamespace(PRN=i.prn, acquiredFreq=i.acquired_freq, codePhase=i.code_phase, status="T") for i in inits]
    t0 = time.perf_counter()
    ref, cdone, aborted = CO.track_l1ca(iq, ch, Sc)
    t_cpu = time.perf_counter() - t0
    cpu_samples = float(np.sum(np.ceil((Sc.codeLength - ref["remCodePhase"]) / (ref["codeFreq"] / fs))))
    cpu_msps = cpu_samples / t_cpu / 1e6
    # Parity at identical descriptors: the CPU loop's own recorded per-epoch state (tracking.m:212-216,249,277,314,332) replayed
    # through the GPU correlator, all 12 x cpu_epochs blocks, against the CPU loop's sums.
    cj = W.Job("cpu", W.PACKAGES["GPS_L1CA"], Sc, main["sats"], eng, params=main["job"].params, inits=inits, phase0=[0] * nch)
    W.keep_records(cj, {k: v for k, v in ref.items()})
    cb, _ = W.replay_blocks(cj)
    got = eng.correlate(cb)[:, 0, :]
    want = np.stack([ref[f].T.reshape(-1) for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L")], axis=1)
    dev = float(np.max(np.abs(got - want))) / (2.0 * 18000 * 28.0)
    same = ref["absoluteSample"] == fields["absoluteSample"][:, :cpu_epochs]
    n_same = [int(np.argmin(r)) if not r.all() else cpu_epochs for r in same]
    # variant (i): NumPy float64, whole-array operations per epoch in the .m's order (the closest stand-in for MATLAB's vector engine)
    np_epochs = min(args.numpy_epochs, cpu_epochs)
    Sn = P.initSettings()
    Sn.msToProcess = np_epochs
    t0 = time.perf_counter()
    nref = O.tracking_l1ca(iq, ch, Sn)
    t_np = time.perf_counter() - t0
    np_samples = float(sum(np.sum(np.ceil((Sn.codeLength - t.remCodePhase) / (t.codeFreq / fs))) for t in nref))
    np_msps = np_samples / t_np / 1e6
    agree = float(max(np.max(np.abs(nref[k].I_P - ref["I_P"][k][:np_epochs])) for k in range(nch)))
    base = {
        "value": round(max(cpu_msps, np_msps) / nch, 3), "unit": "IF Msamples/s (all channels active)", "cores": 1, "kind": "port",
        "host_cpu": _cpu_model(),
        "sample": f"oracle/gnss_oracle.c (float64 restatement of tracking.m:133-368, gcc -O3), closed loop, {nch} channels x {cpu_epochs} epochs "
                  f"of the same record ({t_cpu:.1f} s of CPU); MATLAB-equivalent CPU restatement, not MATLAB",
        "corr_msps": round(cpu_msps, 2), "x_realtime": round(cpu_msps / nch / (fs / 1e6), 4),
        "numpy_variant": {"value": round(np_msps / nch, 3), "corr_msps": round(np_msps, 2), "x_realtime": round(np_msps / nch / (fs / 1e6), 4), "cores": 1,
                          "sample": f"oracle/gnss_oracle.py tracking_l1ca (NumPy float64, whole-array operations per epoch), {nch} channels x {np_epochs} "
                                    f"epochs ({t_np:.1f} s); max |I_P| difference from the C variant {agree:.2e}"},
        "faster_variant": "C" if cpu_msps >= np_msps else "NumPy",
        "gpu_replay_of_cpu_state_max_dev": dev, "closed_loops_cut_the_same_blocks_for_epochs": int(min(n_same)),
    }
    spots = {}
    for jobs in extra_jobs.values():
        for j in jobs:
            spots[j.name] = oracle_spot_check(P, W, O, j)
    return base, spots


def acq_cpu_leg(P, packages, per_kind=3):
    """The acquisition half's CPU baseline (the reference times its own searches: BDS/B1I/include/acquisition.m:77,195
    `timeVec`, GPS_L1CA/include/postProcessing.m:100).  The float64 restatement of acquisition.m (oracle/gnss_oracle.py, NumPy /
    pocketfft, ONE core) on the SAME record as the package's timed GPU search (tests/golden/ref_acq_<pkg>_default.npz's scene), for a
    bounded sample of the PRN list - `per_kind` PRNs the reference detects and `per_kind` it does not, each timed on its own (a
    detected PRN also runs the fine stage) - and scaled to the whole list by the two kinds' means.  Every sampled PRN's codePhase /
    carrFreq must equal the fixture's: the search that is timed is a search that is right."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_scenes as RS
    from oracle import gnss_oracle as O

    def l1ca(rec, S):
        x = rec.astype(np.float64)
        return O.acquisition_l1ca(x[0::2] + 1j * x[1::2], S)
    legs = {"GPS_L1CA": ("acquisition_l1ca (GPS_L1CA/include/acquisition.m:113-260: per PRN 29 bins x 20 hops x (fft + ifft) of 36 000 points, fine stage)", l1ca),
            "BDS_B1I": ("acquisition_b1i (BDS/B1I/include/acquisition.m:34-176: circshift search, 2 carrier shifts x 2 blocks x bins of 72 000 points)", O.acquisition_b1i)}
    out = {}
    for name, (what, fn) in legs.items():
        sc = next(s for s in RS.DEFAULT_ACQ_SCENES if s.name == name + "_default")
        z = np.load(os.path.join(ROOT, "tests", "golden", f"ref_acq_{sc.name}.npz"))
        S, rec = RS.acq_inputs(P, sc)
        full = [int(p) for p in S.acqSatelliteList]
        det = [p for p in full if z["f_carrFreq"][p - 1] != 0]
        und = [p for p in full if z["f_carrFreq"][p - 1] == 0]
        times, equal = {True: [], False: []}, True
        for p in det[:per_kind] + und[:per_kind]:
            S.acqSatelliteList = [p]
            t0 = time.perf_counter()
            a = fn(rec, S)
            times[p in det].append(time.perf_counter() - t0)
            equal = equal and a.codePhase[p - 1] == z["f_codePhase"][p - 1] and a.carrFreq[p - 1] == z["f_carrFreq"][p - 1]
        mean = {k: (sum(v) / len(v) if v else 0.0) for k, v in times.items()}
        cpu_ms = 1e3 * (mean[True] * len(det) + mean[False] * len(und))
        gpu_ms = (packages or {}).get(name, {}).get("ms")
        out[name] = {"ms": round(cpu_ms, 1), "cores": 1, "kind": "port", "unit": "ms per default-size search (whole PRN list)",
                     "ms_per_prn_detected": round(mean[True] * 1e3, 1), "ms_per_prn_not_detected": round(mean[False] * 1e3, 1),
                     "sample": f"oracle/gnss_oracle.py {what}, NumPy float64 on one core: {len(times[True])} detected + {len(times[False])} other PRNs of {len(full)} "
                               f"timed one by one ({sum(times[True]) + sum(times[False]):.1f} s of CPU) on the fixture's record, scaled to {len(det)} + {len(und)} PRNs",
                     "sample_short": f"oracle (NumPy f64) on {len(times[True]) + len(times[False])} of {len(full)} PRNs of the same record, scaled",
                     "sampled_prns_equal_to_the_references_acquisition_m": bool(equal), "gpu_ms": gpu_ms,
                     "gpu_over_cpu": round(cpu_ms / gpu_ms, 1) if gpu_ms else None}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=60.0, help="length of the IF record of the main workload")
    ap.add_argument("--channels", type=int, default=12)
    ap.add_argument("--no-sweep", action="store_true", help="skip closed_loop_sweep (closed loops of 12 ... 192 GPS L1 C/A and 16 ... 64 GPS L5 channels on one GPU)")
    ap.add_argument("--sweep-seconds", type=float, default=10.0, help="record length of the closed-loop channel sweep")
    ap.add_argument("--config", choices=["all", "l1ca", "mix", "sweep"], default=None,
                    help="all (default at N = 1): main line + the other BASELINE configs; l1ca (default at N > 1): main line only; mix: configs[4] as the line")
    ap.add_argument("--prewarm-ms", type=float, default=40.0, help="untimed replay launches before the W warm-up steps, until this many milliseconds have passed (device clocks up after the idle phases); 0 = none")
    ap.add_argument("--cfg-seconds", type=float, default=60.0, help="record length of configs 3 (8 x E1 CBOC at 18 Msps: 15 000 four-millisecond epochs per channel) and 4 (L5 + B2a at 50 Msps)")
    ap.add_argument("--mix-seconds", type=float, default=10.0, help="record length of every band of the mix")
    ap.add_argument("--cpu-epochs", type=int, default=4000, help="epochs per channel timed on the C CPU baseline (4000: ~13 s of one core)")
    ap.add_argument("--numpy-epochs", type=int, default=250, help="epochs per channel of the NumPy CPU variant (~5 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-int16", action="store_true")
    ap.add_argument("--no-acq-packages", action="store_true", help="skip acquisition.packages (the twelve default-size searches checked against their fixtures)")
    ap.add_argument("--no-side-by-side", action="store_true", help="skip replay.side_by_side (a band's packages replayed together on their streams): under the "
                    "kernel-trace profiler the rows of a kernel then hold the one-after-the-other launches only")
    ap.add_argument("--no-handover", action="store_true", help="N > 1: every rank synthesises its own copy of the record(s) instead of receiving them from the source rank (A/B of the exchange step)")
    ap.add_argument("--detail", default=os.environ.get("GC_BENCH_DETAIL", DETAIL_DEFAULT),
                    help="file that receives the full result (every leg, every rank); the printed line is its summary. '' = none")
    ap.add_argument("--spot-check", action="store_true", help="--config mix: every rank checks a few replayed blocks of each of its jobs against the float64 oracle (CPU)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    # More ranks than devices (a 1-GPU box asked for --gpus 8): said at once, as the ONE JSON line, before anything is launched or
    # imported - the library's own device count needs neither torch nor a context.  GC_BENCH_DEVICE pins every rank to one device
    # on purpose (functional tests of the N > 1 path on one GPU).
    if args.gpus > 1 and "GC_BENCH_DEVICE" not in os.environ:
        try:
            from cu_sdr_collection_amd import device_count
            seen = device_count()
        except Exception as e:                      # noqa: BLE001 - no library, no driver: the ranks will say so themselves
            seen = None
            print(f"bench.py: device count unavailable ({e!r})", file=sys.stderr, flush=True)
        if seen is not None and seen < args.gpus:
            if int(os.environ.get("RANK", "0")) == 0:
                print(_error_line(args.gpus, args.steps, args.warmup, f"--gpus {args.gpus} but this process sees {seen} device(s) (gc_device_count); "
                                  "one rank per GPU needs that many (GC_BENCH_DEVICE=<d> pins every rank to one device for functional tests)",
                                  n_devices_seen=seen), flush=True)
            raise SystemExit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus, args.steps, args.warmup))     # no external launcher: one process per GPU, started here
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         "they must agree (n_gpus in the result line is the number of ranks that ran)")
    # one GPU per rank; GC_BENCH_DEVICE pins every rank to one device (functional test of the N > 1 path on a 1-GPU box)
    device = int(os.environ.get("GC_BENCH_DEVICE", local_rank))
    if rank == 0 and world > 1 and "GC_RANK_LAUNCHER" not in os.environ:       # (launch_ranks prints the error line itself)
        # torch.distributed.run (and sharding.launch_ranks) terminate the surviving ranks when one dies: rank 0 still owes the
        # caller ONE JSON line, and says in it that it was stopped from outside
        import signal

        def _terminated(signum, frame):
            print(_error_line(world, args.steps, args.warmup, f"rank 0 received signal {signum} from the launcher before the result line "
                              "(another rank failed or the launcher timed out)"), flush=True)
            os._exit(143)
        signal.signal(signal.SIGTERM, _terminated)
    if rank != 0:
        try:
            os.dup2(2, 1)                   # stdout carries rank 0's ONE line; the other ranks' (and their libraries' C stdio) go to stderr
        except OSError:
            pass
    R = Ranks(rank, world, device)
    config = args.config or ("all" if world == 1 else "l1ca")

    import bench_workloads as W
    import cu_sdr_collection_amd as P

    if config == "mix":
        result = run_mix(P, W, args, R, device)
        _finish(result, args, rank, [R.close])
        return

    result, main_ctx = run_l1ca(P, W, args, R, device)
    result["acquisition"] = run_acquisition(P, main_ctx["eng"], main_ctx["scene"], R)
    if rank == 0 and config == "all" and not args.no_acq_packages:
        result["acquisition"]["packages"] = run_acquisition_packages(P, device)
    extra_jobs = {}
    if config == "all" and world == 1:
        cfgs = {}
        # configs[2]: Galileo E1-C CBOC(6,1,1/11) x 8 (the replica as BASELINE words it; the reference's package tracks BOC(1,1))
        cfgs["galileo_e1c_cboc_x8"], extra_jobs["cboc"] = run_band_jobs(P, W, "config 3", device, [("GAL_E1C_CBOC", 8)], args.cfg_seconds, 18e6, 20e3, 3003,
                                                                          args.steps, args.warmup, side_by_side=not args.no_side_by_side)
        # configs[3]: GPS L5 + BDS B2a, pilot + data arms, 16 channels, 50 Msps - two packages on ONE record
        cfgs["l5_b2a_x16_50msps_int8"], extra_jobs["l5b2a"] = run_band_jobs(P, W, "config 4", device, [("GPS_L5C", 8), ("BDS_B2a", 8)], args.cfg_seconds, 50e6, 20e3,
                                                                             4004, args.steps, args.warmup, side_by_side=not args.no_side_by_side)
        if not args.no_int16:
            cfgs["l5_b2a_x16_50msps_int16"], extra_jobs["l5b2a_int16"] = run_band_jobs(P, W, "config 4 (int16 record)", device, [("GPS_L5C", 8), ("BDS_B2a", 8)],
                                                                                       args.cfg_seconds, 50e6, 20e3, 4004, max(2, args.steps // 4), 1, dtype=np.int16, side_by_side=not args.no_side_by_side)
        # configs[4], one GPU's share: 8 channels of three packages on the L1-band record
        cfgs["mix_share_l1_band_x8"], extra_jobs["mix"] = run_band_jobs(P, W, "config 5 (one GPU's share)", device, [("GPS_L1CA", 3), ("GAL_E1C", 3), ("BDS_B1C_NB", 2)],
                                                                         args.mix_seconds, 18e6, 20e3, 5005, args.steps, args.warmup, side_by_side=not args.no_side_by_side)
        result["configs"] = cfgs
    if rank == 0 and world == 1 and (config == "sweep" or (config == "all" and not args.no_sweep)):
        result["closed_loop_sweep"] = run_closed_loop_sweep(P, W, device, args.sweep_seconds)
    if rank == 0 and world == 1 and not args.no_cpu:
        base, spots = cpu_leg(P, W, args, main_ctx, extra_jobs)
        result["cpu_baseline"] = base
        if spots:
            result["oracle_spot_checks_max_dev_rel_sum_abs_x"] = spots
        if config == "all" and not args.no_acq_packages:
            result["acquisition"]["cpu_baseline"] = acq_cpu_leg(P, result["acquisition"].get("packages"))
    _finish(result, args, rank, [main_ctx["eng"].close, R.close])


if __name__ == "__main__":
    main()
