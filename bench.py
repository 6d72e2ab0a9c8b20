#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native BNN-PYNQ runtime.

Metric (BASELINE.json): images/sec (whole node), CNV-W1A1 on CIFAR-10-shaped
(32x32x3 uint8) batches.  A "step" is one pass of the hot path (all nine stages
of cnvW1A1, image bytes in HBM -> class index in HBM) over one batch of
synthetic images per GPU.  Work per GPU is fixed as N grows ("weak" scaling);
images are independent, so ranks exchange nothing on the data path: the only
collective is the one-time RCCL broadcast of the packed parameter blob.

    python bench.py                          # 1 GPU, defaults finish in ~1-2 min
    python bench.py --gpus N --steps K --warmup W      # starts its own N ranks (one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # or under a launcher: same ranks, same line

Everything the timed region runs goes through the product's C ABI
(include/bnn_mi355x.h).  oracle/ is touched only by the cpu_baseline leg (rank
0, N=1): the CPU restatement timed on this host's cores on a bounded sample,
which doubles as a parity check of the GPU classes on that sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
torch = gl = mg = None  # bound by _rank_imports(): the self-launching parent (below) imports neither torch nor the product


def _rank_imports():
    """torch and the product binding, in a process that is a rank (or the single-GPU run).  torch first: one HIP
    runtime per process."""
    global torch, gl, mg
    import torch as _torch
    import gpu_lib as _gl                 # ctypes binding of the product C ABI
    from bnn import multigpu as _mg       # packed-blob broadcast, shard helpers
    torch, gl, mg = _torch, _gl, _mg


METRIC = "images/sec (whole node) CNV-W1A1 CIFAR-10-shape batch"


# ---- `python3 bench.py --gpus N` started plainly (no torchrun around it): this process becomes a launcher ----------
# It makes NO GPU call and imports neither torch nor the product library: it counts the devices in a short-lived
# child, starts one fresh child process per rank with the environment torch.distributed.run would give it
# (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relays what they print -- rank 0 prints the
# JSON line -- and exits with the first non-zero return code.  Whatever goes wrong before a rank prints the line, the
# launcher prints a JSON line itself ({"error": ...}), so a failed N-rank start is a record, not silence.
def visible_gpus():
    """number of HIP devices a child process sees (the launcher itself never touches the GPU)"""
    import subprocess
    code = "import torch,sys; sys.stdout.write(str(torch.cuda.device_count()))"
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except Exception:
        return 0


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_commands(n, argv, port, python=None, script=None):
    """[(argv, env-additions)] of the N rank processes: this script again with the same arguments"""
    python = python or sys.executable
    script = script or os.path.abspath(__file__)
    out = []
    for r in range(n):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "GROUP_RANK": "0",
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "BNN_BENCH_SELF_LAUNCHED": "1"}
        out.append(([python, "-u", script] + list(argv), env))
    return out


def error_line(a, msg, **extra):
    d = {"metric": METRIC, "error": msg, "value": None, "unit": "images/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup}
    d.update(extra)
    print(json.dumps(d), flush=True)


def self_launch(a, argv):
    import subprocess
    import threading
    need = 1 if a.rehearse_gloo else a.gpus
    have = visible_gpus()
    if have < need:
        error_line(a, "bench.py --gpus %d needs %d visible GPU(s), this machine shows %d" % (a.gpus, need, have), n_gpus_visible=have)
        return 3
    lib = os.path.join(ROOT, "bnn-pynq_amd", "bnn", "libraries", "mi355x", "python_sw-%s-mi355x.so" % a.network)
    if not os.path.exists(lib):   # a checkout without the built libraries: build once, before the ranks race for it
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bnn-pynq_amd")], check=True, stdout=sys.stderr)
    procs, seen_json, lock = [], [False], threading.Lock()

    def relay(rank, pipe):
        for line in pipe:
            with lock:
                if line.startswith("{") and '"metric"' in line:
                    seen_json[0] = True
                    sys.stdout.write(line)
                    sys.stdout.flush()
                else:               # library prints ("Setting network weights ...") of every rank: keep stdout for the line
                    sys.stderr.write("[rank %d] %s" % (rank, line))
    threads = []
    for r, (cmd, env) in enumerate(rank_commands(a.gpus, argv, free_port())):
        p = subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=relay, args=(r, p.stdout), daemon=True)
        t.start()
        threads.append(t)
    rc, failed = 0, None
    live = set(range(a.gpus))
    deadline = None
    give_up = time.time() + a.launch_timeout      # ranks that never come back (a rendezvous that hangs): a record, not silence
    timed_out = False
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc, failed = c, r
                deadline = time.time() + 30.0     # the others hang in a collective without their peer: give them 30 s
        if live and time.time() > give_up and not timed_out:
            timed_out, deadline = True, time.time()
            if rc == 0:
                rc, failed = 124, min(live)
        if deadline and time.time() > deadline:
            for r in live:                        # exactly the processes started above, nothing by pattern
                procs[r].kill()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    if timed_out and not seen_json[0]:
        error_line(a, "the %d ranks did not finish within %d s (rank %d still running): killed" % (a.gpus, a.launch_timeout, failed), returncode=rc)
    elif rc != 0 and not seen_json[0]:
        error_line(a, "rank %d of %d exited with code %d before the result line was printed" % (failed, a.gpus, rc), returncode=rc)
    elif rc == 0 and not seen_json[0]:
        error_line(a, "all %d ranks exited 0 but none printed a result line" % a.gpus)
        rc = 4
    return rc


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)

# algorithmic bytes per image of the fused path: 3072 in + 4 out
ALG_BYTES = {"cnv": 3072 + 4, "lfc": 784 + 4}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer-pipe issue floor (measured in profiles/r01_microbench*.txt): the best schedule found
# for a wave64 logic-op + v_bcnt pair (32 synapses per lane) -- the pair followed by one s_nop 0 -- sustains
# 6.3 SIMD-cycles (6.25..6.40 over runs; v_bitop3 pairs 6.6); v_dot4c and every other VOP3 / SGPR-operand
# integer op 4.2; 1024 SIMDs; 2.38 GHz held under this load.
PAIR_CYC, SLOT_CYC, N_SIMD, CLK_HZ = 6.3, 4.2, 1024, 2.38e9
WORD_MACS = {"cnv": 905216, "lfc": 47104}          # 64-bit word-MACs per image, layers 1..
PAIRS_PER_WORD = {"W1A1": 2, "W1A2": 2, "W2A2": 4}  # (logic op + v_bcnt) pairs per 64-bit word-MAC
# measured (profiles/r01_microbench9_nop_cadence.txt): xor+bcnt 6.3, bitop3+bcnt 6.6, the W2A2 quad as two
# such pairs 12.7 per 32 synapses
PAIR_CYC_OF = {"W1A1": PAIR_CYC, "W1A2": 6.6, "W2A2": 6.35}


def issue_floor_cycles(network, pair_cycles=None):
    """SIMD-cycles per image if the integer pipe issued nothing but the unavoidable instructions (pair_cycles: cycles per
    (logic op, v_bcnt) pair measured in this run instead of the constant)"""
    kind, prec = network[:3], network[3:]
    cyc = WORD_MACS[kind] * PAIRS_PER_WORD[prec] * (pair_cycles if pair_cycles else PAIR_CYC_OF[prec])
    if kind == "cnv":
        # layer 0 runs on the matrix pipe (k_conv0_tile); what stays on the integer pipe per tile (one output row: 30 live
        # pixel lanes of 32) and lane: one v_alignbit per result (32), one v_alignbyte per 3-tap run (5), shifts / merge /
        # store and addressing (~14): 51 as built; 30 row tiles of 64 lanes per image, + quantising every input byte once
        # (768 dwords x 7)
        # (2-bit activations: a second threshold = 32 more results per tile and lane)
        cyc += (30 * 64 * (51 + (32 if prec.endswith("A2") else 0)) + 768 * 7) * SLOT_CYC
    return cyc / 64.0                                # 64 lanes per wave instruction


def calibrate_issue(settle_ms=150):
    """The integer-pipe issue ceiling measured on THIS device, outside every timed region (tools/issue_probe.hip): SIMD-cycles per
    (logic op, v_bcnt) pair in the cadence the product kernels are built to, for the three arithmetic forms, at 4 and 8 waves per
    SIMD, and the clock the device holds under that load (s_memtime / s_memrealtime).  The constants above (round-1 microbenchmarks,
    another box) stay as the `peak` every round has quoted; `peak_measured` / `frac_measured` next to them use these figures."""
    lib = os.path.join(ROOT, "tools", "libissue_probe.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                        os.path.join(ROOT, "tools", "issue_probe.hip"), "-o", lib], check=True)
    P = C.CDLL(lib)
    P.issue_probe_run.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 4
    out = {"source": "tools/issue_probe.hip on this device, outside the timed regions: 16 pairs per asm statement, one s_nop 0 behind each pair, "
                     "4 accumulator chains, SGPR weight, ~1.7 ms launches after 150 ms of the same load.  pair_cycles_measured = the launch's "
                     "duration (HIP events) x the clock / pairs issued per SIMD, the better of 4 and 8 waves per SIMD -- the way the round-1 "
                     "constants were taken; pair_cycles_in_kernel = s_memtime around the loop of the median block at 4 waves per SIMD (every "
                     "block resident) / pairs per SIMD: the loop alone, without the launch; clock = s_memtime / s_memrealtime x 100 MHz",
           "pair_cycles_measured": {}, "pair_cycles_in_kernel": {}, "clock_ghz_measured": {}, "by_waves_per_simd": {}}
    for prec, pat in (("W1A1", 0), ("W1A2", 1), ("W2A2", 2)):
        rows = {}
        for w in (4, 8):
            wall, kern, mhz, ms = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
            if P.issue_probe_run(pat, w, settle_ms if w == 4 else 30, C.byref(wall), C.byref(kern), C.byref(mhz), C.byref(ms)) != 0:
                return {"error": "issue_probe_run(%d, %d) failed" % (pat, w)}
            # (in-kernel only where every block is resident together: the 8-waves-per-SIMD grid ran as two rounds of four on
            # every box so far -- its launch takes twice a block's own time -- so a block's cycles do not cover its SIMD's work)
            rows[str(w)] = {"pair_cycles_wall": round(wall.value, 3), "pair_cycles_in_kernel": round(kern.value, 3) if w == 4 else None,
                            "clock_ghz": round(mhz.value / 1e3, 4), "launch_ms": round(ms.value, 4)}
        best = min(rows.values(), key=lambda r: r["pair_cycles_wall"])
        out["pair_cycles_measured"][prec] = best["pair_cycles_wall"]
        out["pair_cycles_in_kernel"][prec] = rows["4"]["pair_cycles_in_kernel"]
        out["clock_ghz_measured"][prec] = best["clock_ghz"]
        out["by_waves_per_simd"][prec] = rows
    return out


def valu_measured(network, rate_per_gpu, cal):
    """`peak_measured` / `frac_measured` of one network from calibrate_issue()'s figures (None when the probe failed)"""
    if not cal or "error" in cal:
        return {}
    prec = network[3:]
    cyc = issue_floor_cycles(network, pair_cycles=cal["pair_cycles_measured"][prec])
    clk = cal["clock_ghz_measured"][prec] * 1e9
    peak = N_SIMD * clk / cyc
    # the same with the loop's in-kernel rate (no launch in it): the hardware's own issue rate for this instruction mix
    peak_k = N_SIMD * clk / issue_floor_cycles(network, pair_cycles=cal["pair_cycles_in_kernel"][prec])
    return {"peak_measured": round(peak, 1), "frac_measured": round(rate_per_gpu / peak, 4),
            "pair_cycles_measured": cal["pair_cycles_measured"][prec], "clock_ghz_measured": cal["clock_ghz_measured"][prec],
            "peak_in_kernel_rate": round(peak_k, 1), "frac_of_in_kernel_rate": round(rate_per_gpu / peak_k, 4),
            "pair_cycles_in_kernel": cal["pair_cycles_in_kernel"][prec]}


def write_input_file(f, imgs, is_cnv):
    """the reference's input formats: CIFAR-10 binary records (label byte + 3072) / an MNIST idx3 file"""
    n = imgs.shape[0]
    if is_cnv:
        rec = np.empty((n, 3073), np.uint8)
        rec[:, 0] = 1
        rec[:, 1:] = imgs
        f.write(rec.tobytes())
    else:
        f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
    f.flush()


def measure_host_paths(network, dataset, n, dev, device_index, reps=9):
    """The reference's own entry points at a given call size: `inference_multiple(path)` on a file in the page cache (what
    classify_cifars / classify_mnists run: /root/reference/bnn/bnn.py:306-309,370-373 -- the reference's published numbers
    are this call on a 10 000-record test-set file) and `bnn_mi355x_inference_buffer` on a pageable host array, whole calls
    by the wall clock (best and median of `reps`), next to the resident rate of the same images.  Classes of ALL n images
    checked against the CPU restatement, outside the timings."""
    import tempfile
    import oracle_lib as ol
    is_cnv = network.startswith("cnv")
    isz = 3072 if is_cnv else 784
    L = gl.load(network)
    if L.bnn_mi355x_set_device(device_index) != 0:
        return {"error": L.bnn_mi355x_last_error().decode()}
    L.load_parameters(gl.param_dir(dataset, network).encode())
    err = L.bnn_mi355x_last_error().decode()
    if err:
        return {"error": err}
    imgs = np.random.default_rng(4).integers(0, 256, (n, isz), dtype=np.uint8)
    d = torch.from_numpy(imgs).to(dev)
    cls = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    assert L.bnn_mi355x_reserve(n) == 0

    def resident():
        if L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, stream.cuda_stream) != 0:
            raise RuntimeError(L.bnn_mi355x_last_error().decode())
        torch.cuda.synchronize()
    for _ in range(3):
        resident()
    res = []
    for _ in range(reps):
        t = time.perf_counter()
        resident()
        res.append(time.perf_counter() - t)
    usec, cnt = C.c_float(0), C.c_int(0)
    buf, fil = [], []
    got_b = got_f = None
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)                       # the ABI prints its two lines per call, like the reference
    try:
        for _ in range(reps + 1):
            t = time.perf_counter()
            p = L.bnn_mi355x_inference_buffer(imgs.ctypes.data, n, 10, C.byref(usec), 0)
            buf.append(time.perf_counter() - t)
            if not p:
                return {"error": L.bnn_mi355x_last_error().decode()}
            got_b = np.ctypeslib.as_array(p, (n,)).copy()
            L.free_results(p)
        with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
            write_input_file(f, imgs, is_cnv)
            for _ in range(reps + 1):
                t = time.perf_counter()
                p = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), C.byref(usec), 0)
                fil.append(time.perf_counter() - t)
                if not p or cnt.value != n:
                    return {"error": L.bnn_mi355x_last_error().decode()}
                got_f = np.ctypeslib.as_array(p, (n,)).copy()
                L.free_results(p)
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    want = ol.Oracle(network, ol.param_dir(dataset, network)).classes_batched(imgs, 10, host_cores())
    same = bool((got_b == want).all() and (got_f == want).all() and (cls.cpu().numpy() == want).all())
    L.deinit()

    def fig(ts):
        ts = sorted(ts[1:])                   # (the first call sizes buffers)
        return {"best_ms": round(ts[0] * 1e3, 4), "median_ms": round(ts[len(ts) // 2] * 1e3, 4), "value": round(n / ts[0], 1),
                "value_median": round(n / ts[len(ts) // 2], 1), "unit": "images/s"}
    res.sort()
    return {"workload": "%s, %d synthetic images per call: file in the page cache -> classes (inference_multiple) / pageable host array -> "
                        "classes (bnn_mi355x_inference_buffer); whole calls by the wall clock, %d calls each" % (network, n, reps),
            "file_abi": fig(fil), "buffer": fig(buf),
            "resident": {"best_ms": round(res[0] * 1e3, 4), "value": round(n / res[0], 1), "unit": "images/s",
                         "note": "bnn_mi355x_inference_device + synchronize, the same images in HBM"},
            "usecPerImage_reported": round(float(usec.value), 5), "classes_equal_oracle": same, "checked_images": n}


def measure_config(network, dataset, batch, dev, device_index, steps, warmup, check=2048, cal=None):
    """One more single-GPU BASELINE config through the same device-pointer entry point: images/s over `steps`
    timed calls (inputs resident in HBM), HBM- and integer-issue fractions, and the classes of a bounded sample
    compared with the CPU restatement (outside the timed region)."""
    import oracle_lib as ol
    is_cnv = network.startswith("cnv")
    isz = 3072 if is_cnv else 784
    L = gl.load(network)
    if L.bnn_mi355x_set_device(device_index) != 0:   # (a library already bound to this ordinal answers 0)
        return {"error": L.bnn_mi355x_last_error().decode()}
    L.load_parameters(gl.param_dir(dataset, network).encode())
    err = L.bnn_mi355x_last_error().decode()
    if err:
        return {"error": err}
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    imgs = torch.randint(0, 256, (batch, isz), dtype=torch.uint8, device=dev, generator=g)
    classes = torch.zeros(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    assert L.bnn_mi355x_reserve(batch) == 0

    def step():
        if L.bnn_mi355x_inference_device(imgs.data_ptr(), batch, 10, classes.data_ptr(), None, None, stream.cuda_stream) != 0:
            raise RuntimeError(L.bnn_mi355x_last_error().decode())

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # short steps (an LFC batch is 55-600 us): `warmup` of them end before the clocks have settled -- keep the GPU busy for
    # 100 ms before the timed region (tools/batch_sweep.py, which takes the best of four timings, measured 5 % more than
    # this function did with 5 warm-up steps)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.1:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rate = batch / dt
    k = min(check, batch)
    o = ol.Oracle(network, ol.param_dir(dataset, network))
    same = bool((classes[:k].cpu().numpy() == o.classes_batched(imgs[:k].cpu().numpy(), 10, host_cores())).all())
    ceiling = N_SIMD * CLK_HZ / issue_floor_cycles(network)
    alg = ALG_BYTES["cnv" if is_cnv else "lfc"]
    L.deinit()
    return {"workload": "%s, %d synthetic images per step, inputs resident in HBM" % (network, batch), "value": round(rate, 1),
            "unit": "images/s", "us_per_step": round(dt * 1e6, 2), "steps": steps,
            "roofline": {"bound": "hbm", "achieved": round(alg * rate / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg * rate / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_image": alg},
            "valu": dict({"achieved": round(rate, 1), "peak": round(ceiling, 1), "unit": "images/s per GPU", "frac": round(rate / ceiling, 4)},
                         **valu_measured(network, rate, cal)),
            "classes_equal_oracle": same, "checked_images": k}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (1 s of timed device work at the default batch: long enough for a coarse GPU-busy sampler to see it)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=131072, help="images per GPU per step")
    ap.add_argument("--network", default="cnvW1A1")
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target length of the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and integer-pipe-first-layer side measurements")
    ap.add_argument("--dist-world1", action="store_true",
                    help="run the multi-GPU code path (process group, broadcast, gathers, self-check) with a world of ONE rank over "
                         "the real backend: what a one-GPU box can exercise of the RCCL calls")
    ap.add_argument("--launch-timeout", type=int, default=1500,
                    help="self-launched multi-rank run: seconds after which ranks that have not finished are killed and a JSON "
                         "error line is printed")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank dry run on ONE GPU: gloo instead of RCCL, every rank computes on cuda:0 "
                         "(exercises launch/broadcast/shard/timing code where only one GPU is available)")
    return ap.parse_args(argv)


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))    # no GPU call, no torch, no product library in this process
    _rank_imports()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world                            # started by a launcher: its world size is the truth
    is_cnv = a.network.startswith("cnv")
    dataset = a.dataset or ("cifar10" if is_cnv else "mnist")
    ncls = 10
    isz = 3072 if is_cnv else 784

    if not torch.cuda.is_available():
        if rank == 0:
            error_line(a, "bench.py needs a GPU: the product has no CPU path", n_gpus_visible=0)
        sys.exit(3)
    if not a.rehearse_gloo and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        if rank == 0:
            error_line(a, "%d ranks on this node but only %d GPU(s) visible" % (world, torch.cuda.device_count()),
                       n_gpus_visible=torch.cuda.device_count())
        sys.exit(3)
    if a.rehearse_gloo:
        local_rank = 0
    local_rank %= max(torch.cuda.device_count(), 1)   # a launcher may expose one device per rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.dist_world1
    if use_dist:
        import torch.distributed as dist
        if a.dist_world1 and "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.update(RANK="0", WORLD_SIZE="1")
        if a.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    if not os.path.exists(gl.lib_path(a.network)) and rank == 0:
        import subprocess  # a checkout without the built libraries: build them once (hipcc is in the image)
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bnn-pynq_amd")], check=True)
    if use_dist:
        dist.barrier()
    L = gl.load(a.network)
    assert L.bnn_mi355x_set_device(local_rank) == 0

    # ---- parameters: rank 0 packs the reference's param files, everyone else gets the blob over RCCL
    pdir = gl.param_dir(dataset, a.network)
    if not use_dist:
        L.load_parameters(pdir.encode())
        err = L.bnn_mi355x_last_error().decode()
        if err:
            sys.exit(err)
    else:
        # rank 0 reads + repacks the files; the ~210 KB blob goes to the other GPUs over RCCL/xGMI:
        # the one collective of the whole job
        # (= mg.distribute_params, taken apart here to time the collective on its own)
        blob = mg.pack_params(L, pdir) if rank == 0 else None       # host only: read + repack the reference's files
        dist.barrier()
        tb = time.perf_counter()
        buf = mg.broadcast_blob(L, blob, 0, None if a.rehearse_gloo else dev)
        if not a.rehearse_gloo:
            torch.cuda.synchronize()
        broadcast_ms = (time.perf_counter() - tb) * 1e3
        mg.import_blob(L, buf)                                       # straight from HBM (RCCL) / from host memory (gloo)

    # ---- synthetic batch, resident in HBM before the timed region starts
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    imgs = torch.randint(0, 256, (a.batch, isz), dtype=torch.uint8, device=dev, generator=g)
    classes = torch.zeros(a.batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    assert L.bnn_mi355x_reserve(a.batch) == 0

    def step():
        rc = L.bnn_mi355x_inference_device(imgs.data_ptr(), a.batch, ncls, classes.data_ptr(), None, None,
                                           stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(L.bnn_mi355x_last_error().decode())

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    tw = time.perf_counter()
    for _ in range(a.warmup):
        step()
    barrier()
    # the LFC nets' steps take 0.06-0.7 ms: W of them are over before the clocks have settled -- keep the GPU busy until
    # 100 ms of untimed work have passed (decided by the network, not by a timing: every rank takes the same branch)
    if not is_cnv:
        while time.perf_counter() - tw < 0.1:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
        barrier()
    # HIP events around every stage, on the stream the kernels run on, inside the timed region -- where the dispatch policy IS
    # the staged form (the CNV nets above 32 768 images: the default 131 072).  Elsewhere the library may
    # run a network as ONE launch (k_lfc_block_s, k_lfc_fused, k_cnv_tail), which per-stage events would switch off: `value`
    # is then timed on the shipped policy without events, and the per-stage breakdown comes from a second pass of the same K
    # steps with events (roofline.stage_times_source says which).
    # Since late round 3 a CNV pass of 16 384 images and more FORKS over two compute lanes (runtime.hip, bnn_mi355x_inference_device:
    # the halves' kernels overlap, 131 072 images 10.28 -> 10.16 ms); stage events would time overlapping kernels, so profiling
    # keeps one lane -- and the events move to the second pass for these sizes too, unless BNN_MI355X_LANES=1 (the profile
    # runs under rocprofv3: one lane, events in the region, kernel names and durations that match the stage table).
    forks = is_cnv and a.batch >= 16384 and os.environ.get("BNN_MI355X_LANES") != "1"
    events_in_region = is_cnv and a.batch > 32768 and not forks   # (lfcW1A1 is one k_lfc_block_s launch up to 131 072 images, lfcW1A2 up to 2 048)
    L.bnn_mi355x_profile(1 if events_in_region else 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    second_pass_ms = None
    if not events_in_region:
        L.bnn_mi355x_profile(1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        second_pass_ms = (time.perf_counter() - t1) / a.steps * 1e3
    stage_ms = (C.c_float * 16)()
    nchunks = C.c_int(0)
    nst = L.bnn_mi355x_profile_read(stage_ms, 16, C.byref(nchunks))
    L.bnn_mi355x_profile(0)
    multi = None
    if use_dist:
        cdev = "cpu" if a.rehearse_gloo else dev
        own_elapsed = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # ---- self-validation of the N-rank run (all of it outside the timed region): every rank reports its own
        # rate, how long its side of the broadcast took and the CRC-32 of the parameter bytes its GPU holds; the
        # first 2048 images and classes of every rank go to rank 0, which classifies them with the CPU
        # restatement from the parameter FILES (independent of the broadcast blob)
        k = min(2048, a.batch)
        mine = torch.tensor([a.batch * a.steps / own_elapsed, broadcast_ms, float(L.bnn_mi355x_params_crc())], dtype=torch.float64, device=cdev)
        stats = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(stats, mine)
        samp_i, samp_c = imgs[:k].to(cdev).contiguous(), classes[:k].to(cdev).contiguous()
        # (all_gather, RCCL's native collective, rather than gather, which the NCCL backend emulates with grouped send/recv:
        # 6 MB per rank, after the timed region)
        gi = [torch.zeros_like(samp_i) for _ in range(world)]
        gc = [torch.zeros_like(samp_c) for _ in range(world)]
        dist.all_gather(gi, samp_i)
        dist.all_gather(gc, samp_c)
        if rank == 0:
            crcs = ["%08x" % int(x[2].item()) for x in stats]
            multi = {"ranks": world, "backend": "gloo (rehearsal: every rank on cuda:0)" if a.rehearse_gloo else "nccl (RCCL)",
                     "per_rank_images_per_s": [round(float(x[0].item()), 1) for x in stats],
                     "broadcast_ms": [round(float(x[1].item()), 3) for x in stats],
                     "params_crc32": crcs, "params_identical_on_all_ranks": len(set(crcs)) == 1 and crcs[0] != "00000000",
                     "collectives_on_the_data_path": 0}
            if world <= 8:
                import oracle_lib as ol
                o = ol.Oracle(a.network, ol.param_dir(dataset, a.network))
                ok = [bool((gc[r].cpu().numpy() == o.classes_batched(gi[r].cpu().numpy(), ncls, host_cores())).all()) for r in range(world)]
                multi["first_%d_classes_of_every_rank_equal_oracle" % k] = ok
                if not all(ok) or not multi["params_identical_on_all_ranks"]:
                    print(json.dumps({"error": "multi-GPU self-check failed", "multi_gpu": multi}))
                    dist.destroy_process_group()
                    sys.exit("PARITY FAILURE in the %d-rank run" % world)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    total_images = a.batch * a.steps * world
    value = total_images / elapsed
    # ---- roofline: whole fused path priced at its algorithmic bytes, over the device time of all stages
    launches = max(nchunks.value, 1)
    per_stage = [stage_ms[i] / a.steps for i in range(nst)]          # ms per step (all chunks of a step)
    imgs_per_launch = a.batch * a.steps / launches
    names = [L.bnn_mi355x_stage_name(i).decode() for i in range(nst)]
    dom = int(np.argmax(per_stage))
    dev_ms = sum(per_stage)
    alg = ALG_BYTES["cnv" if is_cnv else "lfc"]
    achieved = alg * a.batch / (dev_ms * 1e-3) / 1e9
    traffic, traffic_source, traffic_stages = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf)).get(a.network, {})
            traffic_source = ("profiles/traffic.json: FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per image from the builder's "
                              "two rocprofv3 --pmc passes (%s), scaled to this launch; not re-measured in this run" % t.get("source", "profiles/"))
            # measured HBM bytes per image (PMC passes, profiles/) x the images of one launch
            traffic = int((t["fetch_bytes_per_image_x2"] + t["write_bytes_per_image"]) * imgs_per_launch)
            # per stage, in launch order: where the bytes above the 3 076 algorithmic ones go (the maps between the stages,
            # each written once and read once: no fusion)
            traffic_stages = {names[i] if i < len(names) else st["kernel"]: {"kernel": st["kernel"], "fetch_x2": st["fetch_bytes_per_image_x2"],
                                                                              "write": st["write_bytes_per_image"]}
                              for i, st in enumerate(t.get("stages", []))} or None
        except Exception:
            traffic = None
    # the dominant kernel on its own: bytes that one stage must move per image (input map + output map)
    own_bytes = {"k_conv0 (L0)": 3072 + 7200, "k_quad L1+pool": 7200 + 1568, "k_quad L2": 1568 + 2304, "k_quad L3+pool": 2304 + 400,
                 "k_vec L0": 104 + 128, "k_vec L1": 256, "k_vec L2": 256}
    planes = 2 if a.network.endswith("A2") else 1
    dom_alg = own_bytes.get(names[dom])
    dominant = {"name": names[dom], "ms_per_launch": round(per_stage[dom] * a.steps / launches, 4)}
    if dom_alg:
        dom_alg *= planes
        dominant.update({"algorithmic_bytes_per_image": dom_alg,
                         "achieved": round(dom_alg * a.batch / (per_stage[dom] * 1e-3) / 1e9, 2), "unit": "GB/s",
                         "frac": round(dom_alg * a.batch / (per_stage[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
    # what HBM really moved (the PMC passes' bytes per image x this launch) over the time of a step: "rocprof HBM GB/s against the
    # chip's peak" in the north-star's words; `achieved` prices the path at its ALGORITHMIC bytes
    hbm_measured = None
    if traffic:
        gbs = traffic / (elapsed / a.steps) / 1e9 * (a.batch / imgs_per_launch)
        hbm_measured = {"gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 5), "unit": "GB/s",
                        "note": "roofline.traffic (stored PMC passes, scaled to the images of a step) / ms_per_step"}
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source, "hbm_measured": hbm_measured,
                "traffic_bytes_per_image_by_stage": traffic_stages,
                "kernel": "all %d stages of one batch (dominant: %s, %.1f%% of device time)" % (
                    nst, names[dom], 100.0 * per_stage[dom] / dev_ms),
                "algorithmic_bytes_per_image": alg, "images_per_launch": int(imgs_per_launch),
                "device_ms_per_step": round(dev_ms, 4), "dominant_kernel": dominant,
                "stage_times_source": "HIP events around every stage inside the timed region" if events_in_region else
                "second pass of the same steps with HIP events (staged form, one compute lane); `value` is timed without them on the shipped "
                "dispatch policy, which at this size " + ("forks the pass over two compute lanes (the halves' kernels overlap: ms_per_step is "
                "below the sum of the stages)" if forks else "may be a single launch"),
                "second_pass_ms_per_step": None if second_pass_ms is None else round(second_pass_ms, 4),
                "stages_ms": {names[i]: round(per_stage[i], 4) for i in range(nst)}}
    out = {"metric": METRIC if a.network == "cnvW1A1"
           else "images/sec (whole node) %s batch" % a.network,
           "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u1" if a.network.endswith("A1") else "u2", "dtype_note": "bit-packed XNOR/AND + popcount on the integer VALU; the int8 first layer of the CNV nets on v_mfma_i32_32x32x32_i8", "data": "synthetic",
           "config": {"workload": "%s, %d synthetic %s images per GPU per step, inputs resident in HBM, params %s/%s"
                      % (a.network, a.batch, "32x32x3 uint8" if is_cnv else "28x28 uint8", dataset, a.network),
                      "images_per_gpu_per_step": a.batch, "parallelism": "dp%d (batch shards, no data-path collective)" % world},
           "roofline": roofline}
    if multi:
        out["multi_gpu"] = multi
    floor_cyc = issue_floor_cycles(a.network)
    ceiling = N_SIMD * CLK_HZ / floor_cyc            # images/s per GPU at the issue floor
    per_gpu = a.batch / (dev_ms * 1e-3) if events_in_region else value / world
    out["valu"] = {"bound": "integer-pipe issue (v_xor/v_bitop3 + v_bcnt pairs, v_dot4c)", "achieved": round(per_gpu, 1),
                   "peak": round(ceiling, 1), "unit": "images/s per GPU", "frac": round(per_gpu / ceiling, 4),
                   "simd_cycles_per_image_floor": round(floor_cyc, 1), "clock_ghz": CLK_HZ / 1e9,
                   "pair_cycles": PAIR_CYC_OF[a.network[3:]],
                   "note": "the path is bound by integer VALU issue, not HBM: this is the meaningful ceiling (DESIGN.md 5).  `peak` / `frac` use the "
                           "constants every round has quoted (round-1 microbenchmarks on another box); `peak_measured` / `frac_measured` the "
                           "pair rate and clock tools/issue_probe.hip measured on this device in this run (valu.calibration)"}
    # ---- the issue ceiling measured on this device, in this run, outside the timed region
    cal = None
    if world == 1 and not a.no_extras:
        cal = calibrate_issue()
        out["valu"]["calibration"] = cal
        out["valu"].update(valu_measured(a.network, per_gpu, cal))

    # ---- secondary figures (single GPU only, outside the timed region of `value`)
    if world == 1 and not a.no_extras:
        # (a) PCIe-inclusive: the same batch from pageable HOST memory to int32 classes in host memory
        host_imgs = imgs.cpu().numpy()
        usec = C.c_float(0)
        best = None
        for _ in range(5):
            t1 = time.perf_counter()
            p = L.bnn_mi355x_inference_buffer(host_imgs.ctypes.data, a.batch, ncls, C.byref(usec), 0)
            dt = time.perf_counter() - t1
            if not p:
                sys.exit(L.bnn_mi355x_last_error().decode())
            L.free_results(p)
            best = dt if best is None else min(best, dt)
        out["pcie_inclusive"] = {"value": round(a.batch / best, 1), "unit": "images/s",
                                 "note": "host buffer -> classes in host memory, H2D double-buffered against the stages (DESIGN.md 8); best of 5 calls"}
        # (a') the reference's own entry point: inference_multiple(path) on a file of the same images (page cache)
        import tempfile
        with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
            if is_cnv:
                rec = np.empty((a.batch, 3073), np.uint8)
                rec[:, 0] = 1
                rec[:, 1:] = host_imgs
                f.write(rec.tobytes())
                del rec
            else:
                f.write((0x803).to_bytes(4, "big") + a.batch.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + host_imgs.tobytes())
            f.flush()
            cnt, best = C.c_int(0), None
            for _ in range(6):     # (the first call sizes the record buffers in HBM; best of the rest)
                t1 = time.perf_counter()
                p = L.inference_multiple(f.name.encode(), ncls, C.byref(cnt), C.byref(usec), 0)
                dt = time.perf_counter() - t1
                if not p or cnt.value != a.batch:
                    sys.exit(L.bnn_mi355x_last_error().decode())
                L.free_results(p)
                best = dt if best is None else min(best, dt)
        out["file_abi_inclusive"] = {"value": round(a.batch / best, 1), "unit": "images/s",
                                     "note": "inference_multiple(path): file in the page cache -> classes, streamed to HBM chunk by chunk (DESIGN.md 8); best of 5 calls"}
        del host_imgs
        # (b) the same run with the int8 first layer on the integer pipe (v_dot4c) instead of the matrix pipe
        if is_cnv:
            size = L.bnn_mi355x_export_params(None, 0)
            blob = np.zeros(size, np.uint8)
            L.bnn_mi355x_export_params(blob.ctypes.data, size)
            os.environ["BNN_MI355X_L0"] = "valu"
            L.bnn_mi355x_import_params(blob.ctypes.data, size)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            out["first_layer_on_integer_pipe"] = {"value": round(5 * a.batch / (time.perf_counter() - t1), 1), "unit": "images/s",
                                                  "note": "BNN_MI355X_L0=valu: no MFMA anywhere (DESIGN.md 5, 'Layer 0 on the matrix pipe')"}
            del os.environ["BNN_MI355X_L0"]
            # (c) pricing the north-star's "no MFMA" rule on the layer that takes 46 % of the time: cnvW1A1 layer 1 as an
            # FP4 implicit GEMM on the matrix cores (k_l1_mfma, bit-exact, NOT the product path), everything else as in `value`
            # (d) the same layer in the north-star's literal wording (k_l1_literal: LDS-staged weight tile, __popcll, shuffle
            # pooling; untuned): the measured price of fetching weights from SGPRs and pooling inside a lane instead
            if a.network == "cnvW1A1":
                import oracle_lib as ol
                ref = ol.Oracle(a.network, ol.param_dir(dataset, a.network)).classes_batched(imgs[:2048].cpu().numpy(), ncls, host_cores())
                for mode, key, note in (
                        ("mfma", "matrix_pipe_l1", "BNN_MI355X_L1=mfma: side figure only, the headline stays on XNOR + popcount (DESIGN.md 5, 'Pricing the rule')"),
                        ("lds", "layer1_lds_popcll_shuffle_form", "BNN_MI355X_L1=lds: layer 1 as the north-star words it (weights staged in LDS, __popcll, "
                                "wavefront-shuffle pooling), untuned; comparison only (DESIGN.md 5, 'Where the product departs from the wording')")):
                    os.environ["BNN_MI355X_L1"] = mode
                    L.bnn_mi355x_import_params(blob.ctypes.data, size)
                    for _ in range(2):
                        step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        step()
                    torch.cuda.synchronize()
                    rate = 5 * a.batch / (time.perf_counter() - t1)
                    L.bnn_mi355x_profile(1)
                    step()
                    torch.cuda.synchronize()
                    sm = (C.c_float * 16)()
                    nc = C.c_int(0)
                    L.bnn_mi355x_profile_read(sm, 16, C.byref(nc))
                    L.bnn_mi355x_profile(0)
                    out[key] = {"value": round(rate, 1), "unit": "images/s", "layer1_ms": round(float(sm[1]), 4),
                                "layer1_ms_xnor_popcount": round(per_stage[1], 4),
                                "classes_equal_oracle": bool((classes[:2048].cpu().numpy() == ref).all()), "note": note}
                    del os.environ["BNN_MI355X_L1"]
            L.bnn_mi355x_import_params(blob.ctypes.data, size)
            step()
            torch.cuda.synchronize()

    # ---- the other single-GPU BASELINE configs (2: LFC-W1A1 at its quoted 10 000-image batch; 4: CNV-W2A2), outside
    # the timed region of `value`: a few ms of GPU time each, classes of a sample checked against the oracle
    if world == 1 and not a.no_extras and a.network == "cnvW1A1":
        out["other_configs"] = {
            "lfcW1A1_10000": measure_config("lfcW1A1", "mnist", 10000, dev, local_rank, 300, 20, cal=cal),
            "lfcW1A1_131072": measure_config("lfcW1A1", "mnist", 131072, dev, local_rank, 40, 5, cal=cal),
            "cnvW2A2_131072": measure_config("cnvW2A2", "cifar10", 131072, dev, local_rank, 8, 2, cal=cal),
            # the two W1A2 overlays, same entry point
            "cnvW1A2_131072": measure_config("cnvW1A2", "cifar10", 131072, dev, local_rank, 8, 2, cal=cal),
            "lfcW1A2_131072": measure_config("lfcW1A2", "mnist", 131072, dev, local_rank, 40, 5, cal=cal),
        }
        for key, st in (("cnvW2A2_131072", "cnvW2A2"), ("cnvW1A2_131072", "cnvW1A2"), ("lfcW1A1_131072", "lfcW1A1"),
                        ("lfcW1A2_131072", "lfcW1A2")):                                   # HBM traffic of these configs (stored PMC passes)
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(st)
                if t:
                    per = t["fetch_bytes_per_image_x2"] + t["write_bytes_per_image"]
                    r_ = out["other_configs"][key]["roofline"]
                    r_["traffic"] = int(per * 131072)
                    r_["traffic_source"] = "profiles/traffic.json (%s), scaled to this launch; not re-measured in this run" % t.get("source", "profiles/")
                    r_["hbm_measured"] = {"gbs": round(per * out["other_configs"][key]["value"] / 1e9, 1), "unit": "GB/s",
                                          "frac": round(per * out["other_configs"][key]["value"] / 1e9 / HBM_PEAK_GBS, 5)}
            except Exception:
                pass
        # the reference's own entry points at the reference's own call size (a 10 000-record test-set file: the call behind every
        # number the reference publishes), and the LFC host paths at the headline batch
        hp = measure_host_paths("cnvW1A1", "cifar10", 10000, dev, local_rank, reps=21)   # (1.2 ms a call: the host's memory system is shared with the node's other tenants, short runs scatter by 5-10 %)
        out["other_configs"]["cnvW1A1_10000_file_abi"] = dict(hp.get("file_abi", {}), resident=hp.get("resident"), workload=hp.get("workload"),
                                                              classes_equal_oracle=hp.get("classes_equal_oracle"), checked_images=hp.get("checked_images"),
                                                              **({"error": hp["error"]} if "error" in hp else {}))
        out["other_configs"]["cnvW1A1_10000_buffer"] = dict(hp.get("buffer", {}), resident=hp.get("resident"), workload=hp.get("workload"),
                                                            classes_equal_oracle=hp.get("classes_equal_oracle"), checked_images=hp.get("checked_images"))
        hp = measure_host_paths("lfcW1A1", "mnist", 10000, dev, local_rank, reps=21)
        out["other_configs"]["lfcW1A1_10000_file_abi"] = dict(hp.get("file_abi", {}), buffer=hp.get("buffer"), resident=hp.get("resident"),
                                                              workload=hp.get("workload"), classes_equal_oracle=hp.get("classes_equal_oracle"),
                                                              checked_images=hp.get("checked_images"), **({"error": hp["error"]} if "error" in hp else {}))
        hp = measure_host_paths("lfcW1A1", "mnist", 131072, dev, local_rank, reps=7)
        out["other_configs"]["lfcW1A1_131072_host_paths"] = dict(file_abi=hp.get("file_abi"), buffer=hp.get("buffer"), resident=hp.get("resident"),
                                                                 workload=hp.get("workload"), classes_equal_oracle=hp.get("classes_equal_oracle"),
                                                                 checked_images=hp.get("checked_images"),
                                                                 note="host paths binarise on the host like the reference (104 B per image over PCIe)",
                                                                 **({"error": hp["error"]} if "error" in hp else {}))
        if not all(v.get("classes_equal_oracle") for v in out["other_configs"].values()):
            print(json.dumps(out))
            sys.exit("PARITY FAILURE in other_configs")

    # ---- CPU baseline: the CPU restatement on this host's cores, bounded sample, same images
    if world == 1 and not a.no_cpu_baseline:
        import oracle_lib as ol
        o = ol.Oracle(a.network, ol.param_dir(dataset, a.network))
        host = imgs[: min(a.batch, 262144)].cpu().numpy()
        cores = host_cores()
        probe = min(256, host.shape[0])
        run = (lambda x: o.scores_fast(x, cores)) if is_cnv else (lambda x: o.words_fast(x, cores))
        t1 = time.perf_counter()
        run(host[:probe])
        rate = probe / (time.perf_counter() - t1)
        sample = int(max(probe, min(host.shape[0], rate * a.cpu_seconds)))
        t1 = time.perf_counter()
        run(host[:sample])                       # the timed CPU leg: compute only, like the GPU's timed region
        dt = time.perf_counter() - t1
        ref = o.classes_batched(host[:sample], ncls, cores)
        got = classes[:sample].cpu().numpy()
        if not (got == ref).all():
            sys.exit("PARITY FAILURE: GPU classes differ from the CPU restatement on the baseline sample")
        out["cpu_baseline"] = {"value": round(sample / dt, 1), "unit": "images/s", "cores": cores, "kind": "port",
                               "sample": "first %d images of the GPU batch, %.1f s, popcount+OpenMP restatement "
                                         "(oracle/), classes equal to the GPU's on all of them" % (sample, dt)}
        # the faithful form (one thread, the scalar structure of the reference's HLS C simulation: the analogue
        # of its published 2.39 img/s CNV / 58 img/s LFC on one ARM core), a few images
        k = 3 if is_cnv else 50
        t1 = time.perf_counter()
        for i in range(k):
            if is_cnv:
                s = o.scores_ref(host[i])
                if int(ol.decode_cnv_batched(s, ncls)) != int(got[i]):
                    sys.exit("PARITY FAILURE: GPU class differs from the faithful scalar restatement")
            else:
                o.word_ref(host[i])
        out["cpu_baseline"]["faithful_single_thread"] = {"value": round(k / (time.perf_counter() - t1), 2), "unit": "images/s",
                                                          "cores": 1, "sample": "%d images" % k}
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
