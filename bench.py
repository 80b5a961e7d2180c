#!/usr/bin/env python3
"""bench.py -- prover trace cycles/sec of the sumcheck + HyperKZG hot path on MI355X (BASELINE.json metric).

A "step" is one proof's worth of hot-path work over one synthetic sha3-shaped trace of T = 2^scale cycles
(jolt_amd/workload.py, DeviceWorkload.step), everything a proof needs inside the timed region:
  prepare  witness promotion + every T-sized derived table (eq / eq+1 / LT expansions, linear-leaf fusions) + members
  commit   the committed columns over the shared 2^(4+scale) commitment grid (HyperKZG: MSMs / sums of bases)
  prove    the cycle-domain relations of stages 2..6b as per-stage batched sumchecks
  open     joint polynomial of the homomorphic batch + ONE HyperKZG opening (fold, 4N-term MSM work, Horner, division)
Resident before the timed region: the witness as 64-bit integer columns and hot indices, the SRS ("inputs in HBM").

    python bench.py                         # N=1, BASELINE configs[2]: T=2^22, sumcheck + HyperKZG commit/open end-to-end
    python bench.py --scale 20 --no-msm     # BASELINE configs[1]: T=2^20, sumcheck bind + round-poly kernels only
    python bench.py --gpus N                # hypercube sharded over N GPUs (weak): re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N    # the same under the caller's launcher

Prints ONE JSON line (rank 0) with `roofline` (bind kernel, HIP-event timed live) and `cpu_baseline` (the oracle's
OpenMP port of the same member mix, bounded sample).  The oracle is used only for that baseline leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (8.0 TB/s spec; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=22, help="log2 of the per-GPU trace length T")
    ap.add_argument("--no-msm", action="store_true", help="sumcheck legs only (BASELINE configs[1]); default: commit + open inside the step")
    ap.add_argument("--stages", choices=["all", "2-6b"], default="all",
                    help="all (default): the step also runs the stage 1 / 2 / 5 operators that are not plain cycle-domain relations -- Spartan outer and product, the sparse "
                         "RAM read-write matrix, instruction read-RAF checking with its 128 address rounds (jolt_amd/stages.py); 2-6b: the round-2 step (comparable with BENCH_r02)")
    ap.add_argument("--witness", choices=["resident", "upload", "upload-pinned", "upload-overlapped"], default="resident",
                    help="resident (default, the contract's `value`): the witness columns sit in HBM before the timed region.  upload: every step starts from the packed "
                         "per-cycle rows in HOST memory -- one H2D copy + device-side column extraction (SURVEY.md section 8 f1) -- a PCIe-inclusive diagnostic, N = 1 only")
    ap.add_argument("--ram-addresses", choices=["uniform", "hotset"], default="uniform",
                    help="address stream of the synthetic RAM / register accesses of the stage operators: uniform over the K words, or hotset (90 %% of the accesses on ~2^10 "
                         "words / 8 registers: BASELINE configs[4]'s btreemap shape, specs/byte-addressable-memory.md:119,127-130)")
    ap.add_argument("--transcript", choices=["blake2b", "keccak", "test"], default="blake2b",
                    help="the Fiat-Shamir transcript every sumcheck / opening of the step draws its challenges from: blake2b = the reference's LegacyBlake2bTranscript "
                         "(what its benchmark profile proves with, crates/jolt-prover/src/profile.rs:69), keccak = its KeccakTranscript, test = the deterministic stand-in of rounds 1-5")
    ap.add_argument("--no-split", action="store_true", help="skip the extra (untimed) steps that attribute the step time to its legs")
    ap.add_argument("--roofline-only", action="store_true", help="only run the bind-kernel roofline loop (rocprof target)")
    ap.add_argument("--roofline-scale", type=int, default=24, help="log2 of the table length for the bind roofline (512 MiB at 24: beyond L2+MALL)")
    ap.add_argument("--roofline-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-upload-rate", action="store_true", help="skip the extra timed steps that measure the PCIe-inclusive rate (value_with_upload)")
    ap.add_argument("--no-msm-roofline", action="store_true", help="skip the extra 2^(scale+4)-term MSM that measures the bucket-sum kernel (roofline_msm)")
    ap.add_argument("--cpu-scale", type=int, default=0, help="log2 T of the CPU baseline sample (0 = sized to ~10-30 s of CPU work)")
    ap.add_argument("--round-exchange", choices=["rccl", "shm", "both"], default="both",
                    help="N > 1: how the per-round partial sums are exchanged.  `value` is timed with RCCL (the collective north_star names) unless "
                         "'shm' is given; 'both' (default) times the other exchange in the same run as well and prints the pair in config.round_exchange_ab")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node (one per GPU,
    rendezvous on 127.0.0.1) and hand its exit code back.  Under the driver's own torchrun command WORLD_SIZE is set and this is skipped."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


def bind_roofline(ctx, ffi, log_n, reps):
    """Dominant kernel = k_bind_low_to_high.  Algorithmic bytes per launch = 96 B per output element
    (32*N read + 16*N written = 48*N for a table of N entries), timed with HIP events on the launch
    stream around `reps` launches, each binding a fresh 2^log_n-entry table with a 125-bit challenge."""
    n = 1 << log_n
    rng = np.random.default_rng(7)
    # a full-width pseudo-random table: eq(.) of a random point, built on the device
    pt = rng.integers(0, 2**64, size=(log_n, 4), dtype=np.uint64)
    pt[:, 3] %= np.uint64(0x30644E72E131A029)
    src = ctx.eq_evals(pt)
    r = pt[0].copy()
    r[0] = 0
    r[1] = 0
    r[3] &= np.uint64((1 << 61) - 1)
    m = ctx.member_lc([src], [[(None, [(ffi.host_fr_from_u64(1), 0)])]], 1, borrow=True)  # BORROW member: bind src -> scratch
    m.finish(r)
    m.reset()
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps):
        m.finish(r)
        m.reset()
    ms = ctx.timer_end() / reps
    bytes_per_launch = 48.0 * n
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9
    m.destroy()
    src.free()
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "bind_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "k_bind_low_to_high<shifted>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": "profiles/bind_traffic.json (rocprofv3 --pmc pass of `bench.py --roofline-only`, FETCH_SIZE / WRITE_SIZE corrected per the guide; not collected in this run)" if traffic is not None else None,
            "table_len": n, "bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": round(ms, 5)}


MAD_NOMINAL_T = 39.3  # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: the lane rate if v_mad_u64_u32 issued like a 4-cycle instruction (it does not: ~6.3 cycles per wave
# instruction per SIMD measured, docs/kernels.md section 3.1).  The bound roofline_msm divides by is MEASURED in the run (jolt_ctx_measure_mad_peak, ~50 ms of a register-only
# multiply-add loop on the same device and stream, right before the profiled MSM); the nominal figure is reported beside it.
MADS_PER_MIXED_ADD = 1467  # limb-form XYZZ mixed addition (8 M + 2 S, one two-product reduction): docs/kernels.md section 3.1


def msm_roofline(ctx, srs, log_n):
    """The step's dominant kernel -- the bucket sums of the fixed-base MSM, k_fx_buckets_ordered: 226 of a step's ~380 ms -- against what bounds it.  ONE MSM of 2^log_n
    uniform full-width scalars over the step's own window tables, alone on the device; the kernel is bracketed by HIP events on the stream it is launched on
    (jolt_msm_profile_buckets) and the launch's mixed additions are counted by the digit pass itself.  Bound: integer multiply-add issue (`v_mad_u64_u32`), not HBM and
    not MFMA -- 1467 multiply-adds per addition; the HBM side (a 4-byte index and a 64-byte affine point gathered per addition) is reported beside it."""
    n = 1 << log_n
    rng = np.random.default_rng(11)
    pt = rng.integers(0, 2**64, size=(log_n, 4), dtype=np.uint64)
    pt[:, 3] %= np.uint64(0x30644E72E131A029)
    scalars = ctx.eq_evals(pt)  # a full-width pseudo-random table, built on the device
    ctx.msm(srs, scalars, n, full_width=True)  # warm: workspace growth, attributes
    peak_rate, peak_ms, peak_launches = ctx.measure_mad_peak(50.0)
    peak_t = peak_rate / 1e12
    ctx.msm_profile_buckets(True)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.msm(srs, scalars, n, full_width=True)
    msm_ms = (time.perf_counter() - t0) * 1e3
    ms, adds = ctx.msm_profile_buckets_last()
    ctx.msm_profile_buckets(False)
    scalars.free()
    adds_s = adds / (ms * 1e-3)
    mads_t = adds_s * MADS_PER_MIXED_ADD / 1e12
    alg_bytes = 68.0 * adds
    return {"bound": "valu-int", "kernel": "k_fx_buckets_ordered_staged", "what": f"bucket sums of one fixed-base MSM, 2^{log_n} uniform 254-bit scalars, window tables of the step's SRS",
            "additions": adds, "avg_launch_ms": round(ms, 3), "msm_ms": round(msm_ms, 3),
            "achieved": round(mads_t, 2), "peak": round(peak_t, 2), "unit": "T v_mad_u64_u32/s", "frac": round(mads_t / peak_t, 4),
            "peak_measured": round(peak_t, 3), "peak_source": f"jolt_ctx_measure_mad_peak in this process, this device: best of {peak_launches} launches ({peak_ms:.1f} ms of kernel time, HIP events "
            "on the context's stream) of a register-only loop of independent v_mad_u64_u32 (8 accumulators per lane, 8 workgroups per CU)",
            "peak_nominal": MAD_NOMINAL_T, "frac_of_nominal": round(mads_t / MAD_NOMINAL_T, 4),
            "additions_per_s": round(adds_s), "fq_mul_per_s": round(adds_s * 10), "mads_per_addition": MADS_PER_MIXED_ADD,
            "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1), "frac_of_peak": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "fetched_over_algorithmic": 1.98, "source": "profiles/r05_bucket_index_staging_ab.txt (FETCH_SIZE pass, corrected per the guide): 134.6 B fetched per addition -- the "
                    "128-byte request a 64-byte point gather costs; round 4's 2.64 x included ~52 B per 4-byte index, removed by staging the lists' indices through LDS"},
            "counters": "profiles/r05_pmc_bucket.txt (SQ pass, 3 waves per SIMD at 152 VGPRs): a wave executes a VALU instruction in 34.4 % of its resident cycles, waits to issue in 52 %, waits on memory in 12.6 %",
            "note": "peak = the chip-wide v_mad_u64_u32 rate measured in this run; the kernel's other ~800 instructions per addition share the issue slots, which is why 1.0 is out of reach"}


PUBLISHED_REFERENCE = {  # what the reference itself publishes (whole prover, Dory PCS, CPU; NOT this path alone, NOT this box)
    "value": 1.5e6, "unit": "cycles/s", "what": "whole Jolt prover (all stages, Dory PCS), AMD Threadripper Pro 7975WX, 32 cores",
    "source": "book/src/how/optimizations/inlines.md:147,151,155", "also": "~500 kHz on a MacBook M4 Max, 16 cores (same file :149-150)"}


def cpu_baseline(log_t, srs_dev, with_pcs, with_ext=False, gpu_scale=0):
    """The same step on the host cores through the oracle's OpenMP restatement (kind = "port": the reference is Rust + rayon and
    cannot be built in this image; its arithmetic lives in an un-vendored arkworks fork): per-proof tables, the 11 relations in the
    optimized tier's fused form (skipped s(1), linear combinations folded; the RA columns dense), and -- with_pcs -- the
    commitments on the K x T grid, the joint polynomial and ONE HyperKZG opening.  The MSMs are a signed-digit XYZZ bucket method
    scheduled as one pool of (msm, window, chunk) tasks over all host threads (oracle/g1.c orc_baseline_msm_many), the 64-bit witness
    columns pay only their 5 windows, the one-hot columns are sums of bases (the reference's tier-1 tricks, crates/jolt-dory/src/
    streaming.rs:115-205).  A bounded sample: T = 2^log_t cycles (log_t = 0: 2^20 when a calibration step at 2^16 predicts <= ~40 s
    per step, else the largest T that does), thread count = the faster of all hardware threads and half of them at 2^16."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from jolt_amd import workload as W
    from workload_oracle import OracleExtended, make_table, resolver

    K = 16

    class OracleStageOperators(OracleExtended):
        """the stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators on the host cores: the oracle twins of jolt_amd/stages.py, WITHOUT the from-the-definition address rounds the
        parity tests add (those are checks, not prover work): the 128 read-RAF address rounds run on the library's host state machine, as in the GPU step"""
        @classmethod
        def sampled_direct_rounds(cls, n_vars):
            return set() if n_vars > cls.DIRECT_ADDRESS_ROUNDS_MAX_LOG_T else set(range(128))

    class Sample:
        def __init__(self, scale, pcs=True):
            self.pcs = pcs and with_pcs
            self.scale, self.T = scale, 1 << scale
            self.spec, self.members_spec, gammas = W.build(scale, 2026)
            self.res, self.one, _ = resolver(gammas)
            rng = np.random.default_rng(5)
            self.chal = rng.integers(0, 2**64, size=(scale, 4), dtype=np.uint64)
            self.chal[:, 0] = 0
            self.chal[:, 1] = 0
            self.chal[:, 3] &= np.uint64((1 << 61) - 1)
            prng = np.random.default_rng(6)
            self.idx = np.stack([self.spec[t].data for ms in self.members_spec if ms.uniform is not None for t in ms.tables[1:]])
            self.s_oh, self.s_d = W.rand_fr(self.idx.shape[0], prng), W.rand_fr(2, prng)
            self.point = W.rand_fr(scale + 4, prng)
            if self.pcs:  # the first K * T powers of the device's SRS, converted to affine once (inputs, not timed)
                self.bases = O.baseline_prepare_bases(srs_dev.download(0, K * self.T))
            # The stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators are NOT part of the CPU figure (round-4 review, item 8): their oracle twins are CHECKERS -- dense
            # definitions behind the sparse matrices, brute-force sums behind the read-RAF scans -- not the reference's prefix-suffix / sparse algorithms, so timing
            # them says nothing about the reference's CPU path.  The legs below ARE the reference's algorithms (bind, eq, fused round sums, Pippenger, HyperKZG open).
            self.ext = None

        legs = {"tables": 0.0, "sumchecks": 0.0, "pcs": 0.0, "stage_operators": 0.0}

        def step(self):
            t0 = time.perf_counter()
            tables = {name: make_table(sp) for name, sp in self.spec.items()}  # witness promotion + every derived table
            t1 = time.perf_counter()
            for ms in self.members_spec:
                tabs = [tables[t] for t in ms.tables]
                if ms.split_eq is not None:
                    a, b, pt = ms.split_eq
                    O.baseline_member_sumcheck([O.eq_evals(pt), tabs[a], tabs[b]], [[(None, [(self.one, 0)]), (None, [(self.one, 1)]), (None, [(self.one, 2)])]], 3, self.chal)
                else:
                    O.baseline_member_sumcheck(tabs, self.res.groups(ms.groups), ms.degree, self.chal)
            t2 = time.perf_counter()
            if self.pcs:
                dense = [tables["s6.ram_inc"], tables["s6.rd_inc"]]
                O.baseline_msm_many(self.bases, dense)
                O.baseline_grid_onehot_sums(self.bases, self.idx)
                joint = O.baseline_grid_joint(self.idx, K, self.s_oh, dense, self.s_d)
                O.hyperkzg_open(self.bases, joint, self.point, label=1)
            t3 = time.perf_counter()
            if self.ext is not None:
                self.ext.prove(label=1)
            t4 = time.perf_counter()
            Sample.legs["tables"] += t1 - t0
            Sample.legs["sumchecks"] += t2 - t1
            Sample.legs["pcs"] += t3 - t2
            Sample.legs["stage_operators"] += t4 - t3
            return t4 - t0

    hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    O.baseline_use_parallel_msm(True)
    try:
        # thread count and sample size from calibration steps at 2^16 (the round-2 sample size): all hardware threads vs half of them
        max_grid = (int(np.log2(max(len(srs_dev), 1))) - 4) if with_pcs else 20
        cal_scale = min(16, max_grid)
        cal = Sample(cal_scale)
        best_n, best_c, best_t = None, None, None
        for n in sorted({hw, max(1, hw // 2)}, reverse=True):
            for max_c in (16, 13):  # bucket-window cap: 2^15 XYZZ buckets (4 MiB per thread, L3 / DRAM) against 2^12 (512 KiB, L2) with 23 % more additions
                O.baseline_set_threads(n)
                O.baseline_set_max_window(max_c)
                t = cal.step()
                if best_t is None or t < 0.97 * best_t:
                    best_n, best_c, best_t = n, max_c, t
        O.baseline_set_threads(best_n)
        O.baseline_set_max_window(best_c)
        if log_t <= 0:  # T = 2^20 unless a step there is predicted (linear in T from the calibration step) to exceed about a minute
            log_t = cal_scale
            while log_t < min(22, max_grid) and best_t * (1 << (log_t + 1 - cal_scale)) <= 48.0:  # the contract's bounded sample: ~10-30 s of CPU work per step
                log_t += 1
        smp = cal if log_t == cal_scale else Sample(log_t)
        del cal
        dt, reps = 0.0, 0
        Sample.legs = {k: 0.0 for k in Sample.legs}
        while reps == 0 or (dt < 10.0 and reps < 2):
            dt += smp.step()
            reps += 1
        # the sumcheck legs alone (per-proof tables + the 11 relations) at the GPU line's own T when that fits ~20 s: reported beside the sample, not inside `value`
        legs_at_gpu_t = None
        per_cycle = (Sample.legs["tables"] + Sample.legs["sumchecks"]) / reps / (1 << log_t)
        if gpu_scale and gpu_scale > log_t and per_cycle * (1 << gpu_scale) <= 25.0:
            del smp
            big = Sample(gpu_scale, pcs=False)
            keep = dict(Sample.legs)
            Sample.legs = {k: 0.0 for k in Sample.legs}
            big.step()
            legs_at_gpu_t = {"trace_length": 1 << gpu_scale, "tables_s": round(Sample.legs["tables"], 3), "sumchecks_s": round(Sample.legs["sumchecks"], 3),
                             "cycles_per_s": round((1 << gpu_scale) / (Sample.legs["tables"] + Sample.legs["sumchecks"]), 1)}
            Sample.legs = keep
            del big
    finally:
        O.baseline_use_parallel_msm(False)
        O.baseline_set_max_window(16)
    pcs_note = (f" + commitments of 38 columns on the 2^{log_t + 4} grid (2 MSMs of 64-bit scalars, 36 one-hot sums of bases) + joint polynomial + one HyperKZG opening "
                f"(signed-digit XYZZ bucket MSMs, all level / witness MSMs as one pool of window x chunk tasks; affine bases prepared outside the timed region; "
                f"Horner / RLC passes OpenMP-parallel where the reference's kzg.rs:51-105 is serial)") if with_pcs else ""
    legs_s = {k: round(v / reps, 3) for k, v in Sample.legs.items() if k != "stage_operators"}
    return {"value": round(reps * (1 << log_t) / dt, 1), "unit": "cycles/s", "cores": best_n, "kind": "port",
            "config": {"trace_length": 1 << log_t, "legs": "prepare (per-proof tables) + prove (stage 2-6b sumchecks)" + (" + commit + open" if with_pcs else ""),
                       "seconds_per_step": legs_s, "sumcheck_legs_at_the_gpu_lines_T": legs_at_gpu_t},
            "sample": f"the legs of the step that ARE the reference's algorithms, at T=2^{log_t} on the host cores: "
                      + "per-proof tables + the 11 relations in the optimized tier's fused form (skipped s(1), dense RA columns), all rounds"
                      + " (the stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators are NOT in the CPU figure: their oracle twins are dense-definition checkers, not the reference's "
                        "prefix-suffix / sparse algorithms; `gpu_same_legs` below is the GPU time of exactly the legs timed here)"
                      + f"{pcs_note}; every level commitment of the opening by MSM (the GPU step takes its first two from class sums by linearity); {reps} step(s), C restatement (-O3, 64-bit limbs, ADX) with OpenMP on {best_n} of {hw} host threads "
                      f"(nproc {os.cpu_count()}; the fastest of {hw} / {max(1, hw // 2)} threads x bucket windows capped at 16 / 13 bits at T=2^{cal_scale}: {best_t:.2f} s per step there, cap {best_c}); {dt:.1f}s of CPU work: "
                      f"table builds {Sample.legs['tables']:.1f}s, sumcheck legs {Sample.legs['sumchecks']:.1f}s, commit + open {Sample.legs['pcs']:.1f}s, stage operators {Sample.legs['stage_operators']:.1f}s",
            "published_reference": PUBLISHED_REFERENCE}


def baseline_config(world, scale, with_pcs):
    """which entry of BASELINE.json `configs` a line stands for (the driver's ladder is N = 1, 2, 4, 8 at the default --scale 22: weak scaling, 2^22 cycles per GPU)"""
    total = world << scale
    if not with_pcs:
        return f"configs[1] shape (sumcheck bind + round-poly kernels only, MSM outside the step) at T = 2^{scale}" + ("" if scale == 20 else "; configs[1] itself is --scale 20 --no-msm")
    if world == 1:
        return f"configs[2] (T = 2^{scale}, 1 x MI355X, sumcheck + HyperKZG MSM end-to-end)" + ("" if scale == 22 else "; configs[2] itself is --scale 22")
    lg = total.bit_length() - 1
    if world == 8:
        rel = "configs[3] itself (T = 2^24)" if total == 1 << 24 else f"configs[3]'s shape at {total / (1 << 24):g} x its T = 2^24 (weak scaling keeps configs[2]'s 2^22 cycles per GPU; `--gpus 8 --scale 21` is configs[3]'s own 2^24)"
        return f"{rel}: ONE trace of 2^{lg} cycles, hypercube + MSM sharded over 8 x MI355X, exchanges over RCCL / xGMI"
    return (f"the weak-scaling ladder between configs[2] and configs[3]: ONE trace of {world} x 2^{scale} = 2^{lg} cycles sharded over {world} x MI355X "
            f"(configs[2]'s load per GPU; configs[3] is the 8-GPU rung)")


def main():
    args = parse()
    TR = {"blake2b": 1 << 62, "keccak": 2 << 62, "test": 0}[args.transcript]  # the engine lives in the two top bits of every transcript label (include/jolt_hip.h)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0 and not os.environ.get("JOLT_FORCE_SHARDED"):
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): measuring {world}", file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    sharded = world > 1 or os.environ.get("JOLT_FORCE_SHARDED") == "1"  # the env var exercises the RCCL path with one rank
    # JOLT_BENCH_SHARE_GPU=1: every rank on device 0 with gloo as the rendezvous backend -- exercises the N > 1 code path of this
    # file on a one-GPU box (tests/test_gpu_distributed.py); timings in that mode mean nothing
    share_gpu = os.environ.get("JOLT_BENCH_SHARE_GPU") == "1"
    if sharded:
        import torch
        import torch.distributed as dist
        if share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from jolt_amd import ffi
    from jolt_amd.workload import DeviceWorkload

    if world >= 4 and os.environ.get("JOLT_PCS_SUBTREE", "1") == "0":
        # replicated polynomial arithmetic (A/B): two MSM lanes instead of four from 4 ranks on -- at 8 ranks the opening's polynomials
        # (2^29 coefficients on every rank) take ~110 GB and the rank's window tables 51 GB, and a lane's workspace is 16 GiB
        os.environ.setdefault("JOLT_MSM_LANES", "2")
    ctx = ffi.Context(local_rank if sharded else 0)
    if args.roofline_only:
        print(json.dumps(bind_roofline(ctx, ffi, args.roofline_scale, args.roofline_reps)))
        return

    pcs = None if (args.no_msm or sharded) else "grid"  # (set below for the sharded path once its PCS legs exist)
    if sharded:
        from jolt_amd.distributed import ShardedPcs, ShardedWorkload, make_point_gather
        from jolt_amd.distributed import Collective
        wl = ShardedWorkload(ctx, args.scale, rank, world, dist, force_gather=(world == 1), coll=Collective(dist, world, None) if share_gpu else None)
        pcs_sharded = None
        if not args.no_msm:
            # every rank keeps the raw committed columns of the WHOLE trace resident (inputs: 52 B per cycle), gathered here once
            import torch

            def gather_blocks(local):  # (polys, T_local) or (T_local,) -> the same with world * T_local cycles, blocks in rank order
                t = torch.from_numpy(np.ascontiguousarray(local))
                if not share_gpu:
                    t = t.cuda()
                out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)  # concatenated along dim 0 (both backends accept it)
                dist.all_gather_into_tensor(out, t)
                out = out.reshape((world,) + tuple(t.shape))
                if out.dim() == 3:
                    out = out.permute(1, 0, 2).reshape(t.shape[0], -1)
                else:
                    out = out.reshape(-1)
                return out.cpu().numpy()

            onehot_g = [gather_blocks(a) for a in wl.committed_onehot]
            dense_g = [gather_blocks(d.view(np.int64) if d.dtype == np.uint64 else d).view(d.dtype) for d in wl.committed_dense]
            gp, gfn, guser = make_point_gather(wl.coll, world)
            # block-cyclic term assignment: every rank keeps window tables over ITS 2^(4 + scale) bases at any world size;
            # JOLT_PCS_BLOCK_CYCLIC=0: contiguous term ranges against the full SRS (tables only for world <= 2), for an A/B
            block_cyclic = os.environ.get("JOLT_PCS_BLOCK_CYCLIC", "1") != "0"
            # JOLT_PCS_SUBTREE=0: the opening's polynomial arithmetic replicated on every rank (only its MSMs sharded) instead of the subtree-sharded opening
            subtree = block_cyclic and world >= 2 and world & (world - 1) == 0 and os.environ.get("JOLT_PCS_SUBTREE", "1") != "0"
            pcs_sharded = ShardedPcs(ctx, rank, world, args.scale, onehot_g, dense_g, gp, gfn, guser, fixed_base=(block_cyclic or world <= 2), block_cyclic=block_cyclic,
                                     subtree=subtree)
            pcs = "grid"

        ext = None
        if args.stages == "all":
            # the stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators over ONE trace of world * 2^scale cycles dealt to the ranks in blocks (jolt_amd/stages_sharded.py): additive
            # scans / pushforwards summed over the ranks, sharded cycle-domain batches, replicated K-sized rounds, sparse matrices with local cycle rounds + merged rows
            from jolt_amd.stages_sharded import ShardedExtended
            ext = ShardedExtended(ctx, args.scale, rank, world, wl.coll, tail_log=wl.tail_log, round_exchange=wl.round_exchange)

        def step(label=0):  # the same legs as the N = 1 step (DeviceWorkload.step): prepare, commit, extended operators, prove, open
            wl.prepare()
            if pcs_sharded is not None:
                pcs_sharded.commit()
            if ext is not None:
                ext.prove(label)
            wl.prove(label=label)
            if pcs_sharded is not None:
                pcs_sharded.open(label)
    else:
        wl = DeviceWorkload(ctx, args.scale, pcs=pcs, extended=(args.stages == "all"), ram_addresses=args.ram_addresses, witness_upload={"resident": False, "upload": True, "upload-pinned": "pinned", "upload-overlapped": "overlapped"}[args.witness])
        if os.environ.get("JOLT_FLIP_UPLOAD") == "1":  # diagnostic: a RESIDENT-built workload switched to the overlapped upload before the timed loop (what the value_with_upload leg does)
            wl.witness_upload, wl.witness_pinned, wl.witness_overlapped = True, True, True
        step = wl.step

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
            ctx.synchronize()

    def timed(steps, warmup, base):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; max over the ranks."""
        for i in range(warmup):
            step(label=TR | (base + 1000 + i))
        barrier()
        if sharded:
            for k in _D.TIMINGS:
                _D.TIMINGS[k] = 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            step(label=TR | (base + 2000 + i))
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    exchange_ab = None
    if sharded:
        from jolt_amd import distributed as _D
        shm = wl.round_exchange  # None when the ranks could not agree on a shared-memory segment (or JOLT_ROUND_EXCHANGE says otherwise)
        primary = "shm" if (args.round_exchange == "shm" and shm is not None) else "rccl"
        if args.round_exchange == "both" and shm is not None and world > 1:  # the exchange that is NOT reported as `value`, timed first
            other = "shm" if primary == "rccl" else "rccl"
            wl.round_exchange = shm if other == "shm" else None
            if ext is not None:
                ext.round_exchange = wl.round_exchange
            exchange_ab = {other: round(timed(args.steps, args.warmup, 50000) / args.steps * 1e3, 3)}
        wl.round_exchange = shm if primary == "shm" else None
        if ext is not None:
            ext.round_exchange = wl.round_exchange
    total_cycles_for_upload = 1 << args.scale
    dt = timed(args.steps, args.warmup, 0)
    if exchange_ab is not None:
        exchange_ab[primary] = round(dt / args.steps * 1e3, 3)
    ms_per_step = dt / args.steps * 1e3
    # where the step time goes: two more steps (outside the timed region) with a synchronisation after each leg
    split = None
    if os.environ.get("JOLT_FLIP_UPLOAD") == "after" and not sharded:  # diagnostic: the legs of a step AFTER a resident timed loop, with the overlapped upload switched on
        wl.witness_upload, wl.witness_pinned, wl.witness_overlapped = True, True, True
        args.witness = "upload-overlapped"
        for i in range(3):
            step(label=TR | (90000 + i))
    if not sharded and not args.no_split:
        ext_legs = []
        if wl.ext is not None:
            e, dd = wl.ext, wl.ext.d
            boo = {}  # the booleanity address phase's output: the cycle phase starts from its bound point
            ext_legs = [("spartan_outer", lambda: e.spartan(e.outer_ints, dd["outer_iwa"], dd["outer_iwb"], dd["outer_wa"], dd["outer_wb"], dd["outer_tau"], dd["outer_kernel"],
                                                           e.claims["outer"], 2, TR | 3100)),
                        ("spartan_product", lambda: e.spartan(e.product_ints, e.product_ia, e.product_ib, e.product_fa, e.product_fb, dd["product_tau"], dd["product_kernel"],
                                                             e.claims["product"], 1, TR | 3200)),
                        ("ram_read_write", lambda: e.ram_read_write(TR | 3300)), ("registers_read_write", lambda: e.registers_read_write(TR | 3350)),
                        ("instruction_read_raf", lambda: e.instruction_read_raf(TR | 3400)), ("booleanity_address", lambda: boo.update(e.booleanity_address(TR | 3450))),
                        ("booleanity_cycle", lambda: e.booleanity_cycle(TR | 3460, boo["challenges"][::-1], boo["intermediate"])),
                        ("hamming_weight", lambda: e.hamming_weight(TR | 3470)), ("address_domain", lambda: e.address_domain(TR | 3500))]
        # "opening_hint_background": what is left of the class sums the commit leg began on the background stream once the commit itself has landed -- in the step
        # they run UNDER the stage operators and the sumcheck legs (and slow those down a little: the legs of this split, timed one by one, add up to more than the step)
        hint_leg = [("opening_hint_background", lambda: wl.hint.wait() if getattr(wl, "hint", None) is not None else None)] if pcs else []
        legs = ([("witness_upload", wl.upload_witness)] if args.witness != "resident" else []) + [("prepare", wl.prepare)] + ([("commit", wl.commit)] if pcs else []) + hint_leg + ext_legs + \
               [("prove", lambda: wl.prove(label=TR | 3000))] + \
               ([("stages_1_to_7_as_in_the_step", lambda: wl.prove_stages(label=TR | 3000))] if getattr(wl, "_stage_worker", None) is not None else []) + \
               ([("open", lambda: wl.open(label=TR | 3000))] if pcs else [])
        acc = {k: 0.0 for k, _ in legs}
        reps = 2
        for _ in range(reps):
            for k, fn in legs:
                ctx.synchronize_foreground() if k == "opening_hint_background" else ctx.synchronize()
                t1 = time.perf_counter()
                fn()
                ctx.synchronize_foreground() if k == "commit" else ctx.synchronize()
                acc[k] += time.perf_counter() - t1
        split = {k: round(v / reps * 1e3, 3) for k, v in acc.items()}
    # The PCIe-inclusive rate beside the contract's `value` (never instead of it): the same K steps with every proof starting from packed rows in page-locked HOST memory,
    # the next proof's copy in flight under the current proof's kernels (jolt_rows_upload_begin; a tracer one trace ahead of the prover).  N = 1, resident default run only.
    with_upload = None
    if pcs and not sharded and args.witness == "resident" and not args.no_upload_rate:
        try:
            wl.witness_upload, wl.witness_pinned, wl.witness_overlapped = True, True, True
            dt_up = timed(args.steps, max(3, args.warmup), 70000)  # (the first proof has no copy ahead of it, the second meets a cold pool)
            bpc = wl.witness_bytes_per_cycle()
            # the resident step once more, right behind it: by now the part has run ~30 steps back to back and clocks lower than in the first timed loop, so the upload's
            # cost is the difference to THIS figure, not to `ms_per_step`
            wl.witness_upload = wl.witness_overlapped = False
            dt_res = timed(args.steps, 1, 80000)
            with_upload = {"value": round(total_cycles_for_upload / (dt_up / args.steps), 1), "unit": "cycles/s", "ms_per_step": round(dt_up / args.steps * 1e3, 3),
                           "resident_ms_per_step_right_after": round(dt_res / args.steps * 1e3, 3),
                           "bytes_per_cycle": bpc, "bytes_per_step": bpc << args.scale,
                           "mode": "every step starts from packed per-cycle rows in page-locked host memory (catalogue witness: integer columns + RA chunk addresses; the stage operators' inputs stay "
                                   "resident); one H2D copy per proof on a copy stream, begun when the PREVIOUS proof has extracted its columns, so it runs under that proof's kernels; "
                                   "device-side column extraction inside the step"}
        except Exception as e:  # a diagnostic: never fail the bench on it
            with_upload = {"value": None, "error": str(e)}
        finally:
            wl.witness_upload = wl.witness_overlapped = False
    n_onehot = getattr(wl, "n_onehot", 0) or sum(a.shape[0] for a in getattr(wl, "committed_onehot", []))
    onehot_note = f" of which {n_onehot} are one-hot RA selector columns kept as 1-byte hot indices until their fourth bind" if n_onehot else ""
    total_cycles = (1 << args.scale) * world
    ext_note = ""
    the_ext = ext if sharded else wl.ext
    if the_ext is not None:
        ram = the_ext.d["ram"] if hasattr(the_ext, "d") else {"log_k": the_ext.p["ram_log_k"]}
        bc_log_k = the_ext.d["bytecode"]["log_k"] if hasattr(the_ext, "d") else the_ext.p["bytecode"]["log_k"]
        ext_note = (f"runs the stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators outside the cycle-domain catalogue -- Spartan outer (uni-skip sums off 35 integer columns, Az / Bz, log T + 1 "
                    f"remainder rounds, claimed inputs), Spartan product (the same over the 6 product lanes), the sparse RAM read-write matrix (K = 2^{ram['log_k']}, "
                    f"log T + log K rounds), registers read-write checking (<= 3 cells per cycle over 128 registers, log T + 7 rounds, operand claims) and instruction read-RAF checking end to end over the 42 lookup tables (per address phase the T-scale scans on the device and the 8 rounds over 256-entry prefix / suffix polynomials on the host, 128 address rounds in all, then log T cycle rounds), the booleanity address phase and the Hamming-weight claim reduction (pushforward masses of the 36 RA columns + log K host rounds each), the booleanity cycle phase (the 36 columns lazily bound, log T rounds; N = 1 only), and the address-domain relations -- bytecode read+RAF (five per-stage "
                    f"pushforwards onto the 2^{bc_log_k}-entry bytecode domain + log K rounds, then C * prod ra_i over log T rounds), RAM RAF evaluation and the RAM output check "
                    f"(pushforward / final-memory column over a sorted index of the address column + log K rounds each)"
                    + (f" (ONE trace of {world} x 2^{args.scale} cycles dealt to the ranks in blocks: additive scans and pushforwards summed over the ranks, cycle-domain "
                       f"sumchecks sharded like the catalogue's, K-sized rounds replicated, the sparse matrices' rows merged after the local cycle rounds)" if sharded else "") + " -- ")
    if pcs and sharded:
        what = (f"BASELINE configs[2] sharded over {world} GPU(s): sha3-shaped synthetic trace of {world} x 2^{args.scale} cycles, sumcheck + HyperKZG end-to-end -- "
                f"every step commits the {n_onehot + 2} committed columns on the 2^{pcs_sharded.grid_vars} commitment grid (each rank its block of cycles, one all-gather of "
                f"partial points), {ext_note}proves the stage 2-6b cycle-domain sumchecks hypercube-sharded (11 relations, {wl.n_tables} T-sized tables per rank) and opens the "
                f"joint polynomial (2^{pcs_sharded.grid_vars} coefficients) with ONE HyperKZG opening "
                + (f"sharded over the ranks by index subtree (every rank builds, folds, combines, divides and commits 1/{world} of the polynomial against its own "
                   f"bases and window tables; O(ell) field elements and points exchanged)" if pcs_sharded.subtree else
                   f"whose MSMs are split over the ranks {'block-cyclically (term i belongs to rank (i / 2^%d) mod %d; window tables over the rank own bases)' % (args.scale, world) if pcs_sharded.block else 'by term range'} (polynomial arithmetic replicated)")
                + f"; the raw committed columns of the whole trace (52 B per cycle) are resident on every rank; every step rebuilds the rank's "
                f"per-proof tables (witness promotion, eq blocks, linear-leaf fusions, members) like the N=1 step")
    elif pcs:
        what = (f"BASELINE configs[2]: sha3-shaped synthetic trace, T=2^{args.scale} per GPU, sumcheck + HyperKZG end-to-end -- every step rebuilds the "
                f"per-proof tables (witness promotion, eq / eq+1 / LT expansions, linear-leaf fusions, members), commits the {n_onehot + 2} committed columns "
                f"on the 2^{wl.grid_vars} commitment grid (2 dense MSMs of T 64-bit scalars + {n_onehot} one-hot columns as sums of bases), {ext_note}proves the "
                f"stage 2-6b cycle-domain sumchecks (11 relations, {wl.n_tables} T-sized tables{onehot_note}, degree 2-5) and opens the joint polynomial "
                f"(2^{wl.grid_vars} coefficients) with ONE HyperKZG opening whose first two level commitments are combined from class sums of the one-hot columns' bases instead of "
                f"MSMs over the folded coefficients (all of it -- sums, MSMs, commit and open -- inside the timed region)")
    else:
        what = (f"sha3-shaped synthetic trace, T=2^{args.scale} per GPU: per-proof tables + {ext_note}stages 2-6b cycle-domain sumchecks "
                f"(11 relations, {wl.n_tables} T-sized tables{onehot_note}, degree 2-5), bind + round-poly HIP kernels; MSM not in the timed region "
                f"(BASELINE configs[1] shape)")
    out = {
        "metric": "prover trace cycles/sec (sha3-shaped synthetic trace, sumcheck + HyperKZG hot path)",
        "value": round(total_cycles / (dt / args.steps), 1),
        "unit": "cycles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u256 (BN254 Fr / Fq, 8x u32 Montgomery limbs; integer, bit-exact)",
        "data": "synthetic",
        "config": {"workload": what, "trace_length_per_gpu": 1 << args.scale, "trace_length_total": total_cycles, "baseline_config": baseline_config(world, args.scale, bool(pcs)),
                   "parallelism": f"hypercube sharded over {world} GPU(s)", "ram_addresses": args.ram_addresses,
                   "transcript": {"blake2b": "jolt_transcript::LegacyBlake2bTranscript (DigestTranscript<Blake2b-256>: the transcript of the reference's benchmark profile, crates/jolt-prover/src/profile.rs:69), "
                                             "round polynomials / commitments / evaluations absorbed in the reference's encodings",
                                  "keccak": "jolt_transcript::KeccakTranscript (spongefish Keccak duplex sponge; pinned by the reference's known-answer vector)",
                                  "test": "the deterministic test transcript of rounds 1-5"}[args.transcript]},
    }
    if split is not None:
        out["config"]["ms_per_step_split"] = split
        if "stages_1_to_7_as_in_the_step" in split:
            out["config"]["ms_per_step_split_note"] = ("every operator leg and `prove` are timed ALONE, one after the other; in the step the operators of protocol stage k run on their own "
                                                       "context and host thread beside stage k's batched sumcheck (never across a stage boundary): `stages_1_to_7_as_in_the_step` is that "
                                                       "whole block timed as the step runs it, and replaces the sum of the operator legs + `prove` when the legs are added up")
    if not sharded and args.witness != "resident":
        bpc = wl.witness_bytes_per_cycle()
        out["config"]["witness"] = {"mode": f"upload: every step starts from packed rows in {'page-locked (jolt_host_pinned_alloc)' if args.witness != 'upload' else 'pageable'} host memory{', the next proof copied under the current one (jolt_rows_upload_begin)' if args.witness == 'upload-overlapped' else ''} (PCIe-inclusive; NOT the contract's `value`, which has the inputs resident)",
                                    "bytes_per_cycle": bpc, "bytes_per_step": bpc << args.scale,
                                    "h2d_GBps": round((bpc << args.scale) / (split["witness_upload"] * 1e-3) / 1e9, 2) if split and split.get("witness_upload") and args.witness != "upload-overlapped" else None,  # (overlapped: the copy is not inside the leg)
                                    "note": "catalogue witness only (integer columns + RA chunk addresses); the stage operators' inputs stay resident"}
    if with_upload is not None:
        out["value_with_upload"] = with_upload["value"]
        out["config"]["witness_upload"] = with_upload
    if not sharded:
        out["config"]["device_pool_gib"] = {k.replace("_bytes", ""): round(v / 2**30, 2) for k, v in ctx.memory_stats().items()}
        if not sharded and getattr(wl, "ext_ctx", None) is not None:  # the stage operators' own context
            out["config"]["device_pool_gib"]["stage_operator_context"] = {k.replace("_bytes", ""): round(v / 2**30, 2) for k, v in wl.ext_ctx.memory_stats().items() if "msm" not in k}
    if sharded:
        rounds = "shared-memory exchange of the round sums between the ranks of the node" if wl.round_exchange is not None else "RCCL all-gather of the round sums"
        out["config"]["collective"] = f"{rounds}; {type(wl.coll).__name__}: RCCL all-gather of the 2^tail_log-entry tables"
        # what actually ran (a silent fallback would change what is measured): the exchange of the per-round sums and the communicator
        out["config"]["round_exchange"] = ("shm" if wl.round_exchange is not None else ("rccl" if type(wl.coll).__name__ == "NativeCollective" else "torch.distributed")) \
            + (" (JOLT_ROUND_EXCHANGE=%s)" % os.environ["JOLT_ROUND_EXCHANGE"] if os.environ.get("JOLT_ROUND_EXCHANGE") else "")
        out["config"]["communicator"] = getattr(wl, "communicator_note", type(wl.coll).__name__) + \
            f"; torch.distributed backend {dist.get_backend()} with {dist.get_world_size()} rank(s)"
        if exchange_ab is not None:  # ms per step with either exchange of the round sums, same run, same steps / warmup (`value` is the first key's)
            out["config"]["round_exchange_ab"] = {primary: exchange_ab[primary], **{k: v for k, v in exchange_ab.items() if k != primary}}
        out["config"]["tail_log"] = wl.tail_log
        if pcs_sharded is not None:
            out["config"]["pcs"] = (f"commitments: block-cyclic shares over per-rank compact bases; opening: polynomial sharded by index subtree, per-rank window tables, partial points and O(ell) field elements through {type(wl.coll).__name__}"
                                    if pcs_sharded.subtree else
                                    f"block-cyclic sharded MSMs (block 2^{args.scale} terms) over per-rank compact bases with fixed-base window tables, partial points through {type(wl.coll).__name__}"
                                    if pcs_sharded.block else
                                    f"term-range sharded MSMs, partial points through {type(wl.coll).__name__}; fixed-base window tables {'on' if world <= 2 else 'off (memory)'}")
        out["config"]["ms_per_step_split"] = {k: round(v / args.steps * 1e3, 3) for k, v in _D.TIMINGS.items()}
    if rank == 0:
        out["roofline"] = bind_roofline(ctx, ffi, args.roofline_scale, args.roofline_reps)
        out["roofline"]["note"] = ("measured on a 2^%d-entry table (beyond L2 + the 256 MiB Infinity Cache): an HBM figure; the binds inside the step run on 2^%d-entry "
                                   "tables (%d MiB each) and are served in part from the Infinity Cache" % (args.roofline_scale, args.scale, (32 << args.scale) >> 20))
        if split is not None and "prove" in split:
            # the sumcheck legs as a whole against the HBM roofline (fused bind + evaluate form: 32 m N read + 16 m N written per round, all rounds of a
            # table of N entries = 96 N (1 - 2^-n) bytes), m = the dense T-sized tables the round kernels read (one-hot RA columns and factored-out eq tables excluded)
            def dense_tables(ms):  # what the member's round kernels read at T entries
                if ms.fused is not None:
                    return len(ms.fused[1])  # the fused leaves replace their sources
                if ms.uniform is not None:
                    return 0 if wl._lazy(ms) else len(ms.tables) - 1  # index-encoded one-hot columns; the eq table is never materialised
                return len(ms.tables) - (1 if ms.eq_inner is not None else 0)
            m_dense = sum(dense_tables(ms) for ms in wl.members_spec)
            alg = 96.0 * m_dense * (1 << args.scale) * (1.0 - 2.0 ** -args.scale)
            ach = alg / (split["prove"] * 1e-3) / 1e9
            out["roofline_sumcheck"] = {"bound": "hbm", "what": "stage 2-6b batched sumchecks (leg `prove`), fused bind + round-evaluation form", "achieved": round(ach, 1),
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "dense_tables": m_dense,
                                        "bytes_per_proof": alg, "prove_ms": split["prove"],
                                        "note": "the round kernels are bound by 256-bit multiplies, not by bytes (DESIGN.md section 3.3): this fraction is reported, not a target met"}
        if pcs and not sharded and not args.no_msm_roofline:
            try:
                out["roofline_msm"] = msm_roofline(ctx, wl.srs, args.scale + 4)
            except Exception as e:  # a diagnostic object: never fail the bench on it
                out["roofline_msm"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            try:
                srs_dev = (pcs_sharded.srs if sharded else wl.srs) if pcs else None  # the CPU sample's grid uses a prefix of the same SRS (inputs)
                out["cpu_baseline"] = cpu_baseline(args.cpu_scale, srs_dev, bool(pcs), with_ext=(the_ext is not None), gpu_scale=args.scale)
                at_t = os.path.join(ROOT, "profiles", "r06_cpu_baseline_T22.json")
                if args.cpu_scale == 0 and args.scale == 22 and os.path.exists(at_t):  # the same CPU legs at the GPU line's own T: ~2.5 minutes of CPU work, collected once per round
                    try:
                        rec = json.load(open(at_t))
                        out["cpu_baseline"]["config"]["at_gpu_T"] = {"trace_length": rec["config"]["trace_length"], "value": rec["value"], "cores": rec["cores"],
                                                                      "seconds_per_step": rec["config"]["seconds_per_step"],
                                                                      "source": "profiles/r06_cpu_baseline_T22.json (`bench.py --cpu-scale 22` on a GPU box's host; NOT collected in this run)"}
                    except Exception:
                        pass
                if split is not None:  # the GPU's time for exactly the legs the CPU figure covers (the stage operators are in `value` but not in the CPU sample)
                    same = [k for k in ("prepare", "commit", "opening_hint_background", "prove", "open") if k in split]
                    ms = sum(split[k] for k in same)
                    out["cpu_baseline"]["gpu_same_legs"] = {"legs": same, "ms_per_step": round(ms, 3), "cycles_per_s": round((1 << args.scale) / (ms * 1e-3), 1),
                                                            "ratio_to_cpu": round(((1 << args.scale) / (ms * 1e-3)) / out["cpu_baseline"]["value"], 1) if out["cpu_baseline"].get("value") else None}
            except Exception as e:  # the oracle is optional infrastructure: never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "cycles/s", "cores": None, "kind": "port", "sample": f"unavailable: {e}", "published_reference": PUBLISHED_REFERENCE}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
