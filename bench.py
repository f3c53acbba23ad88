#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: CSR SpMV achieved HBM GB/s (+ CG iterations/s)
on the 5-point 2-D Poisson matrix, n = 3162^2 = 9,998,244 rows per GPU, and the
strong-scaling CG on the 7-point 464^3 grid (C5) that north_star's ">= 6x at 8 GPUs" is quoted on.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process only SPAWNS the N
                                                            rank processes, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: `--spmv-per-step` (default 512) back-to-back
products y = A x on the resident matrix (halo exchange included when N > 1) -- about 50 ms of GPU
work, so that the K timed steps are not a 2 ms window.  Inputs are resident in HBM before the
timed region.

Accounting.  `value` and `roofline.achieved` count the bytes the running kernel
MOVES by construction -- its stored format as it reads it (sgm_mat_footprint: padded slices,
codes, row pointers) + every x entry once + every y entry once -- divided by wall time / by the
HIP-event launch time.  The reference layout's algorithmic bytes (12 nnz + 4 (n+1) + 8 m + 8 n,
SURVEY §8d) divided by the same times are reported SEPARATELY as
`effective_GBs_on_reference_bytes`; that figure may exceed the HBM peak because the kernel reads
a compressed layout, and it is never called "achieved HBM".
N > 1 is weak scaling for the C2 line: every rank owns nx*ny rows of an nx x (N*ny) grid.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
PROFILE_TAG = "r06"


def spmv_bytes(n, m, nnz):
    """Algorithmic bytes of the REFERENCE layout: int32 ptr/node + fp64 val, x, y."""
    return 12 * nnz + 4 * (n + 1) + 8 * m + 8 * n


def csrc_sha1():
    """Fingerprint of the kernel sources; PMC summaries under profiles/ carry the one they were
    collected with, and `traffic` is only quoted when it matches what runs now."""
    h = hashlib.sha1()
    for f in ("sgm_spmv.hip", "sgm_spmv_select.hpp", "sgm_internal.hpp"):          # the SpMV kernels and the header they share
        h.update(open(os.path.join(ROOT, "sigma_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spmv-per-step", type=int, default=512,
                    help="products y = A x per timed step (one step is about 50 ms of GPU work)")
    ap.add_argument("--nx", type=int, default=3162)
    ap.add_argument("--ny", type=int, default=3162)
    ap.add_argument("--cg-steps", type=int, default=300)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the other CSR kernels on the same matrix")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 strong-scaling CG leg")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"],
                    help="workload of the timed SpMV steps.  c2: 5-point 2-D Poisson, nx*ny rows PER GPU (weak "
                         "scaling, the BASELINE metric); c5: 7-point 3-D Laplacian m^3 split in z-slabs (strong)")
    ap.add_argument("--c5-edge", type=int, default=464, help="grid edge m of the C5 grid (m^3 rows)")
    ap.add_argument("--c5-cg-steps", type=int, default=200)
    ap.add_argument("--force-dist", action="store_true",
                    help="take the RCCL row-partition code path even with one rank (testing aid)")
    ap.add_argument("--halo-comm", action="store_true",
                    help="N > 1: a second RCCL communicator for the halo send/recv pairs (A/B switch for halo / all-reduce overlap)")
    ap.add_argument("--no-pcg", action="store_true", help="N = 1: skip the `pcg_time_to_solution` leg (CG vs ILDU(0)-PCG, 1000^2 grid)")
    ap.add_argument("--no-dist-overhead", action="store_true",
                    help="N = 1: skip the `dist_overhead_1rank` leg (CG through the RCCL code path with one rank)")
    ap.add_argument("--no-c5-parts", action="store_true",
                    help="N = 1: skip the `c5_8parts` leg (the 464^3 matrix as 8 in-process z-slabs on this GPU: what ONE of 8 ranks "
                         "has to do per CG iteration, and the speed-up that bounds)")
    ap.add_argument("--c5-parts", type=int, default=8)
    ap.add_argument("--halo-fused", type=int, default=1, choices=[0, 1, 2],
                    help="N > 1: option dist_halo_fused of the CG solves (1: r's boundary rows in one RCCL group with the all-reduce of "
                         "r.r, p's halo formed locally; 2: a group of their own; 0: p exchanged in front of every product)")
    ap.add_argument("--no-ceilings", action="store_true", help="N = 1: skip tools/stream_bench (measured streaming ceilings of this GPU)")
    ap.add_argument("--no-c3", action="store_true", help="N = 1: skip the C3 leg (1-D advection-diffusion n = 1e7: SpMV, BiCGStab, GMRES(30))")
    ap.add_argument("--no-c4", action="store_true", help="N = 1: skip the C4 leg (ELLPACK random digraph, degree 32, n = 5e6: SpMV)")
    ap.add_argument("--c3-n", type=int, default=10_000_000)
    ap.add_argument("--c3-iters", type=int, default=300)
    ap.add_argument("--c4-n", type=int, default=5_000_000)
    ap.add_argument("--pcg-nx", type=int, default=3162, help="N = 1: grid edge of the second `pcg_time_to_solution` row (0 = skip it)")
    ap.add_argument("--stall-s", type=float, default=300.0,
                    help="every rank: seconds without a heartbeat (a phase change of this script, or a beat of the library: "
                         "solver batches, halo posts, all-reduces) after which the rank prints where it is stuck and exits 86")
    ap.add_argument("--deadline-s", type=float, default=1500.0,
                    help="whole-run limit: a rank past it prints where it is and exits 86; with `--gpus N` typed plainly the "
                         "parent also kills its children (by PID) then and prints every rank's last phase")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ #
# Heartbeat + watchdog: ranks that all block in a collective would otherwise run into
# the caller's timeout silently.  Every rank keeps "<dir>/rank<r>.hb" = its current phase; a daemon thread
# looks once a second at (this script's phase counter, the library's own heartbeat -- sgm_heartbeat, readable
# while the main thread is blocked inside the library) and, when nothing has moved for --stall-s or the run is
# past --deadline-s, writes the state, prints it and ends the process with os._exit(86).  Under
# torch.distributed.run that makes the launcher end the other ranks; typed as `python bench.py --gpus N` the
# parent does the same by PID.  No process is ever exec()ed or killed by pattern.
# ------------------------------------------------------------------------------------------ #
class Heartbeat:
    EXIT = 86

    def __init__(self, rank, stall_s, deadline_s, sg=None):
        import threading
        self.rank, self.stall_s, self.deadline_s, self.sg = rank, stall_s, deadline_s, sg
        self.dir = os.environ.get("SGM_BENCH_HB_DIR")
        self.name, self.count, self.t_phase = "start", 0, time.time()
        self.t0 = time.time()
        self.done = False
        self._last_sig, self._last_move = None, time.time()
        self._write()
        self._thr = threading.Thread(target=self._watch, daemon=True)
        self._thr.start()

    def phase(self, name):
        self.name, self.count, self.t_phase = name, self.count + 1, time.time()
        self._write()

    def state(self):
        st = {"rank": self.rank, "phase": self.name, "phase_no": self.count, "in_phase_s": round(time.time() - self.t_phase, 1),
              "elapsed_s": round(time.time() - self.t0, 1)}
        if self.sg is not None:
            try:
                st["library"] = self.sg.heartbeat()
            except Exception as e:       # the watchdog must never be what takes the run down
                st["library"] = {"error": str(e)[:100]}
        return st

    def _write(self, extra=None):
        if not self.dir:
            return
        try:
            st = self.state()
            if extra:
                st.update(extra)
            tmp = os.path.join(self.dir, f"rank{self.rank}.hb.tmp")
            with open(tmp, "w") as f:
                json.dump(st, f)
            os.replace(tmp, os.path.join(self.dir, f"rank{self.rank}.hb"))
        except OSError:
            pass

    def _watch(self):
        while not self.done:
            time.sleep(1.0)
            st = self.state()
            lib = st.get("library") or {}
            sig = (st["phase_no"], lib.get("beats"), lib.get("iteration"))
            now = time.time()
            if sig != self._last_sig:
                self._last_sig, self._last_move = sig, now
            why = None
            if now - self._last_move > self.stall_s:
                why = f"no heartbeat for {now - self._last_move:.0f} s (--stall-s {self.stall_s:g})"
            elif now - self.t0 > self.deadline_s:
                why = f"run past --deadline-s {self.deadline_s:g}"
            if why and not self.done:
                self._write({"stalled": why})
                sys.stderr.write(f"[bench] rank {self.rank} STALLED: {why}; state: {json.dumps(st)}\n")
                sys.stderr.flush()
                os._exit(self.EXIT)

    def finish(self):
        self.done = True
        self.phase("done")


# ------------------------------------------------------------------------------------------ #
# N > 1 typed as `python bench.py --gpus N`: spawn the rank processes.  This parent never
# touches the GPU (no torch.cuda / HIP call, not even an import of torch) and never exec()s.
# ------------------------------------------------------------------------------------------ #
def spawn_ranks(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    # rank 0's line goes through a temporary FILE, not a pipe: nothing a child prints can fill a pipe buffer and block it
    # while this parent waits for it to exit
    cap = tempfile.TemporaryFile()
    hb_dir = tempfile.mkdtemp(prefix="sgm_bench_hb_")
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SGM_BENCH_CHILD": "1", "SGM_BENCH_HB_DIR": hb_dir})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=cap if r == 0 else subprocess.DEVNULL))

    def last_phases():
        out = []
        for r in range(args.gpus):
            try:
                out.append(json.load(open(os.path.join(hb_dir, f"rank{r}.hb"))))
            except (OSError, ValueError):
                out.append({"rank": r, "phase": "(no heartbeat file)"})
        return out

    # if one rank dies the others would wait in a collective for ever: end them (by PID) after a grace period; if ALL of them
    # block (every rank waits in a collective) the children's own watchdogs fire after --stall-s, and this parent is the
    # backstop at --deadline-s (+ a margin so that the children get to report first)
    t_start, grace, killed = time.time(), None, False
    while any(p.poll() is None for p in procs):
        rcs = [p.poll() for p in procs]
        if grace is None and any(rc not in (None, 0) for rc in rcs):
            grace = time.time() + 20.0
        if (grace is not None and time.time() > grace) or time.time() - t_start > args.deadline_s + 15.0:
            if not killed:
                sys.stderr.write("[bench] ending the remaining ranks; last heartbeat of every rank:\n")
                for st in last_phases():
                    sys.stderr.write("[bench]   " + json.dumps(st) + "\n")
            killed = True
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    cap.seek(0)
    out = cap.read().decode()
    cap.close()
    sys.stdout.write(out)
    sys.stdout.flush()
    rcs = [p.returncode for p in procs]
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        sys.stderr.write(f"[bench] rank exit codes: {rcs}\n")
        if not killed:
            for st in last_phases():
                sys.stderr.write("[bench]   " + json.dumps(st) + "\n")
    import shutil
    shutil.rmtree(hb_dir, ignore_errors=True)
    return min(worst, 255)


def main():
    args = parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world != args.gpus:
        if os.environ.get("SGM_BENCH_CHILD"):
            sys.exit("bench.py child started without its rank environment")
        sys.exit(spawn_ranks(args))
    if env_world > 1 and args.gpus != env_world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}")
    worker(args)


# ------------------------------------------------------------------------------------------ #
def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import sigma_amd as sg
    from sigma_amd import problems as P

    # The one JSON line is the ONLY thing this process writes to its stdout: native libraries print there too (librccl
    # writes a version banner to fd 1 when its first communicator comes up), so fd 1 is pointed at stderr for the rest of
    # the run and the line goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SGM_BENCH_SAME_GPU"):      # testing aid: all ranks on device 0 (with the mock transport)
        local_rank = 0
    sg.init(local_rank)                           # fails loudly (SGM_ERR_NO_DEVICE) without a GPU
    hb = Heartbeat(rank, args.stall_s, args.deadline_s, sg)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # every kernel of the library is launched on THIS torch stream, so torch.cuda.Event
    # (HIP events) brackets exactly the launches it is recorded around
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sg.use_torch_stream()
    sg.set_async(True)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # torch.distributed is the CONTROL plane only (bootstrap of the RCCL id, barriers, the
        # max-over-ranks of the timings) and runs over gloo on the host, so that the data
        # plane -- the library's own RCCL communicators (halo send/recv + dot all-reduces on the
        # launch stream) -- are the only RCCL communicators of the process.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        hb.phase("control plane: torch.distributed gloo rendezvous")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def sum_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    comm = None
    if use_dist:
        hb.phase("RCCL bootstrap: unique id broadcast + sgm_comm_init (ncclCommInitRank)")
        uid = [sg.Comm.unique_id() if rank == 0 else None, sg.Comm.unique_id() if (rank == 0 and args.halo_comm) else None]
        dist.broadcast_object_list(uid, src=0)
        comm = sg.Comm(rank, world, uid[0], uid[1])
    # CG's communication pattern on a partition (option dist_halo_fused): 1 = the boundary rows of r in ONE RCCL group with the
    # all-reduce of r.r.  Every rank first posts such a group once (send / recv to itself + an all-reduce over all ranks): a
    # transport that refuses the mixed group says so HERE, and the solves fall back to two separate calls (mode 2).
    halo_mode = 0 if args.halo_comm else args.halo_fused
    group_probe = None
    if use_dist and halo_mode == 1:
        hb.phase("RCCL: probing a group of send / recv + all-reduce (sgm_comm_group_selftest)")
        try:
            got, summed, us = comm.group_selftest()
            ok = got == 42.0 + rank and summed == float(world)
            group_probe = {"ok": bool(ok), "us": us}
        except sg.SigmaError as e:
            ok = False
            group_probe = {"ok": False, "error": str(e)[:200]}
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if t.item() != 1.0:
            halo_mode = 2
            if rank == 0:
                sys.stderr.write(f"[bench] the transport refused a group of send / recv + all-reduce ({group_probe}): CG posts them separately (dist_halo_fused = 2)\n")

    def make_matrix(kind):
        """(A, n_loc, n_glob, i0, nnz, label, host_arrays)"""
        host = None
        hb.phase(f"{kind}: generating this rank's rows + create" + (" (sgm_csr_create_dist: collective)" if use_dist else ""))
        if kind == "c2":
            nx, ny = args.nx, args.ny
            n_loc = nx * ny
            n_glob, i0 = n_loc * world, rank * n_loc
            starts = np.arange(world + 1, dtype=np.int64) * n_loc
            if not use_dist:
                host = P.poisson2d_csr(nx, ny)
                arrays = tuple(torch.from_numpy(a).to(dev) for a in host)
            else:
                arrays = local_rows_poisson2d(nx, ny, world, rank)
            label = f"5-point 2D Poisson CSR, {nx}x{ny} rows per GPU"
        else:
            m = args.c5_edge
            zs = [(m * r) // world for r in range(world + 1)]
            starts = np.array([z * m * m for z in zs], dtype=np.int64)
            i0, n_loc, n_glob = int(starts[rank]), int(starts[rank + 1] - starts[rank]), m ** 3
            arrays = local_rows_laplace3d(m, int(zs[rank]), int(zs[rank + 1]), dev)
            label = f"7-point 3D Laplacian CSR {m}^3 split in z-slabs over {world} GPU(s)"
        nnz = int(arrays[2].numel() if hasattr(arrays[2], "numel") else len(arrays[2]))
        torch.cuda.synchronize()
        if not use_dist:
            A = sg.csr_matrix(n_loc, n_loc, *arrays)
        else:
            A = sg.dist_csr_matrix(comm, starts, *arrays)
        del arrays
        return A, n_loc, n_glob, i0, nnz, label, host

    def vectors(A, n_loc, i0):
        x = torch.zeros(A.x_len if use_dist else n_loc, dtype=torch.float64, device=dev)
        x[:n_loc] = torch.sin(0.001 * torch.arange(i0 + 1, i0 + n_loc + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        return x, y

    def selfcheck(kind, y, n_loc, i0):
        """Every local row of one product against its sum evaluated with torch in stored order (the x entries come from
        the analytic x(i) = sin(0.001 i), so the halo values a rank needs are known to it): bit for bit, on every rank.
        A product that differs must not be reported as a measurement."""
        k = torch.arange(i0, i0 + n_loc, device=dev, dtype=torch.int64)
        if kind == "c2":
            nx, NY = args.nx, args.ny * world
            i, j = k % nx, k // nx
            one = torch.ones_like(k, dtype=torch.bool)
            offs = [(-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 4.0), (1, i < nx - 1, -1.0), (nx, j < NY - 1, -1.0)]
        else:
            m = args.c5_edge
            pl = m * m
            i, j, l = k % m, (k // m) % m, k // pl
            one = torch.ones_like(k, dtype=torch.bool)
            offs = [(-pl, l > 0, -1.0), (-m, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 6.0), (1, i < m - 1, -1.0),
                    (m, j < m - 1, -1.0), (pl, l < m - 1, -1.0)]
        z = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        for o, mask, v in offs:
            xo = torch.sin(0.001 * (k + (o + 1)).to(torch.float64))
            z = torch.where(mask, z + v * xo, z)
        z = 0.0 + z
        ok = bool(torch.equal(z, y[:n_loc]))
        if use_dist:
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item() == 1.0)
        return ok

    def time_kernel(mat, x, y, reps=200, flush=None):
        """Average duration of ONE product, HIP events on the launch stream around every launch.
        flush: a scratch tensor rewritten between launches (cold Infinity Cache / L2)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            if flush is not None:
                flush.add_(1.0)
            a.record()
            mat.matvec(x, y)
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3

    def cg_run(A, n_loc, n_glob, its_cap, profile_phases=False, tag=""):
        hb.phase(f"{tag}: CG, {its_cap} fixed iterations (warm-up solve, timed solve" + (", profiled solve)" if profile_phases else ")"))
        s = sg.cg(1e-300)
        s.set_max_iter(its_cap)
        s.set_option("dist_halo_fused", halo_mode)     # (a second communicator only serves mode 0)
        s.setup(A)
        bvec = torch.full((n_loc,), 1.0 / n_glob, dtype=torch.float64, device=dev)
        u = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        s.solve(A, u, bvec, check=False)       # warm-up
        u.zero_()
        barrier()
        t0 = time.perf_counter()
        s.solve(A, u, bvec, check=False)
        barrier()
        dtc = max_over_ranks(time.perf_counter() - t0)
        its, res2 = s.last_iterations, s.res2
        phases = None
        if profile_phases:
            # a THIRD solve with the library's HIP-event phase timers on (kept out of the timed one: ~14 event records per
            # iteration): where an iteration's time goes on this rank, and the slowest rank's figure per phase
            u.zero_()
            sg.dist_profile(True)
            s.solve(A, u, bvec, check=False)
            pr = sg.dist_profile_read()
            sg.dist_profile(False)
            n_it = max(1, s.last_iterations)
            phases = {nm: {"ms_per_iter_rank0": v["ms"] / n_it, "ms_per_iter_max_over_ranks": max_over_ranks(v["ms"] / n_it),
                           "events_per_iter": v["count"] / n_it} for nm, v in pr.items()}
        s.destroy()
        return its, dtc, res2, phases

    # ---- N = 1: what a plain streaming kernel reaches on THIS GPU today (tools/stream_bench, a child process) --------------
    ceilings = None
    if rank == 0 and world == 1 and not use_dist and not args.no_ceilings and args.workload == "c2":
        hb.phase("stream ceilings: tools/stream_bench at 600 and 7600 MiB")
        ceilings = stream_ceilings()

    # ---- the workload of the timed steps -----------------------------------------------------
    A, n_loc, n_glob, i0, nnz, label, host = make_matrix(args.workload)
    x, y = vectors(A, n_loc, i0)
    alg_bytes_rank = spmv_bytes(n_loc, n_loc, nnz)
    resident_rank, moved_rank = A.footprint()
    inner = max(1, args.spmv_per_step)

    hb.phase(f"{args.workload}: first product + self-check against the stored-order row sums")
    A.matvec(x, y)
    torch.cuda.synchronize()
    check_main = selfcheck(args.workload, y, n_loc, i0)
    if not check_main:
        sys.stderr.write(f"[bench] rank {rank}: the product differs from its row sums in stored order -- not a measurement\n")
    hb.phase(f"{args.workload}: {args.warmup} warm-up steps of {inner} products")
    for _ in range(args.warmup):
        for _ in range(inner):
            A.matvec(x, y)
    hb.phase(f"{args.workload}: barrier before the timed steps")
    barrier()
    hb.phase(f"{args.workload}: {args.steps} timed steps of {inner} products (halo exchange per product when N > 1)")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(inner):
            A.matvec(x, y)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    launches = args.steps * inner
    ms_per_step = 1e3 * dt / args.steps
    moved_all = sum_over_ranks(moved_rank)
    alg_all = sum_over_ranks(alg_bytes_rank)
    value = moved_all * launches / dt / 1e9
    effective = alg_all * launches / dt / 1e9

    # ---- dominant kernel with HIP events on the launch stream -------------------------------
    hb.phase(f"{args.workload}: per-launch HIP-event timing (warm and cold)")
    k_avg = time_kernel(A, x, y)
    main_kernel = A.kernel
    scratch = torch.zeros(512 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)     # 512 MiB
    k_cold = time_kernel(A, x, y, reps=40, flush=scratch)
    achieved = moved_rank / k_avg / 1e9

    def variant_entry(mat, t, t_cold=None):
        _, mv = mat.footprint()
        e = {"kernel": mat.kernel, "avg_launch_ms": 1e3 * t, "moved_bytes_per_launch": mv,
             "GB/s_moved": mv / t / 1e9, "frac_of_hbm_peak": mv / t / 1e9 / HBM_PEAK_GBS,
             "effective_GBs_on_reference_bytes": alg_bytes_rank / t / 1e9}
        if t_cold:
            e["cold_launch_ms"] = 1e3 * t_cold
            e["cold_frac_of_hbm_peak"] = mv / t_cold / 1e9 / HBM_PEAK_GBS
        return e

    variants = {}
    if not use_dist and not args.no_variants and args.workload == "c2":
        hb.phase("c2: the other CSR kernels on the same matrix")
        variants["sliced 4-bit codes, 2 rows per lane (default for rows <= 8 entries / <= 15 offsets)"] = \
            variant_entry(A, k_avg, k_cold)
        for vlabel, opts in (("offset_dict_u8_codes, LDS-staged row-owner kernel (stencil-like matrices with longer rows / more offsets)",
                              {"csr_sliced": 0}),
                             ("int32_columns, row-owner gather (general kernel, rows <= 64 entries)",
                              {"csr_offset_dict": 0, "csr_row_owner": 1}),
                             ("int32_columns, streaming gather (general kernel, any row length)",
                              {"csr_offset_dict": 0, "csr_row_owner": 0, "csr_row_lines": 0})):
            for k, v in opts.items():
                A.set_option(k, v)              # this handle's own options (they are per handle)
            for _ in range(5):
                A.matvec(x, y)
            variants[vlabel] = variant_entry(A, time_kernel(A, x, y, reps=50), time_kernel(A, x, y, reps=20, flush=scratch))
            for k in ("csr_offset_dict", "csr_row_owner", "csr_row_lines", "csr_sliced"):
                A.set_option(k, 1)
        # a handle created WITHOUT the dictionary: short rows take the sliced int32-column kernel
        # (what a matrix with arbitrary columns and rows <= 32 entries of similar length gets; 12 B per slot)
        sg.set_option("csr_offset_dict", 0)
        try:
            A32 = sg.csr_matrix(n_loc, n_loc, A.get("ptr", np.int32), A.get("node", np.int32), A.get("val", np.float64))
        finally:
            sg.set_option("csr_offset_dict", 1)
        for _ in range(5):
            A32.matvec(x, y)
        variants["int32_columns, sliced (general kernel, rows <= 32 entries of similar length)"] = \
            variant_entry(A32, time_kernel(A32, x, y, reps=50), time_kernel(A32, x, y, reps=20, flush=scratch))
        A32.destroy()
    del scratch

    # ---- CG iterations/s (device-resident loop, fixed iteration count) -----------------------
    cg = None
    in_solver_ms = None
    if args.cg_steps > 0:
        its, dtc, res2, cg_phases = cg_run(A, n_loc, n_glob, args.cg_steps, profile_phases=True, tag=args.workload)
        # moved per iteration: the SpMV's bytes + 8 vector passes (q written by the SpMV is counted there;
        # r-update reads r,q writes r; x/p update reads x,p,r writes x,p) = 64 n;
        # SURVEY §8d grades on the fused floor B_csr + 72 n of the REFERENCE layout -- both reported
        moved_it = sum_over_ranks(moved_rank + 64 * n_loc)
        floor_it = sum_over_ranks(alg_bytes_rank + 72 * n_loc)
        cg = {"iters_per_s": its / dtc, "iterations": its, "ms_per_iter": 1e3 * dtc / its,
              "moved_bytes_per_iter": moved_it, "GB/s_moved": moved_it * its / dtc / 1e9,
              "frac_of_hbm_peak": moved_it * its / dtc / 1e9 / (HBM_PEAK_GBS * world),
              "effective_GBs_on_survey_floor": floor_it * its / dtc / 1e9, "final_res2": res2,
              "phases": cg_phases if use_dist else None}
        # the product as it runs INSIDE the solver (every launch after two update passes have gone through the caches),
        # from the library's HIP events around each of its launches
        in_solver_ms = None if use_dist or not cg_phases else cg_phases["interior_rows"]["ms_per_iter_rank0"] / max(1e-9, cg_phases["interior_rows"]["events_per_iter"])

    # ---- N = 1: the fixed cost of the RCCL code path (real librccl, ONE rank) on the same matrix ----------------
    dist_overhead = None
    if rank == 0 and world == 1 and not use_dist and not args.no_dist_overhead and args.cg_steps > 0 and args.workload == "c2":
        hb.phase("c2: CG through the RCCL code path with one rank (dist_overhead_1rank)")
        dist_overhead = dist_overhead_leg(args, sg, torch, dev, cg["iters_per_s"] if cg else None)

    # ---- CPU baseline (rank 0, N = 1): the reference itself on the SAME matrix ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and host is not None:
        hb.phase("cpu_baseline: the reference + the oracle port on the host cores (about 25 s)")
        cpu = cpu_baseline(args, host, n_loc)

    # ---- C1 (BASELINE configs[0], the reference's own CPU-sized case): tridiagonal n = 10,000, CG to 1e-16 ----
    c1 = None
    if rank == 0 and world == 1 and not args.no_cpu:
        hb.phase("c1: tridiagonal n = 10,000, CG to 1e-16 (device, device in the reference's dot order, reference on the host)")
        c1 = c1_leg(sg, P, torch, dev)

    # ---- preconditioned solves, time to solution on the 1000^2 grid (setup + solve): CG, ILDU(0)-PCG in natural and colour order ----
    pcg = None
    if rank == 0 and world == 1 and not args.no_pcg:
        hb.phase("pcg_time_to_solution: CG vs ILDU(0)-PCG (natural / colour order), 1000^2")
        pcg = pcg_leg(sg, P, torch, dev)
        if args.pcg_nx and args.pcg_nx != 1000:
            hb.phase(f"pcg_time_to_solution: the same at {args.pcg_nx}^2 (C2 size)")
            pcg = {"grid_1000": pcg, f"grid_{args.pcg_nx}": pcg_leg(sg, P, torch, dev, nx=args.pcg_nx)}

    kernel_sha = csrc_sha1()
    A.destroy()
    del x, y

    # ---- C5: 7-point 464^3 split in z-slabs over the N GPUs (strong scaling) -------------------
    c5 = None
    if not args.no_c5 and args.workload == "c2":
        A5, n5, n5g, i5, nnz5, label5, _ = make_matrix("c5")
        x5, y5 = vectors(A5, n5, i5)
        hb.phase("c5: first product + self-check")
        A5.matvec(x5, y5)
        torch.cuda.synchronize()
        check_c5 = selfcheck("c5", y5, n5, i5)
        if not check_c5:
            sys.stderr.write(f"[bench] rank {rank}: the C5 product differs from its row sums in stored order\n")
        hb.phase("c5: 3 + 20 products between barriers")
        for _ in range(3):
            A5.matvec(x5, y5)
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            A5.matvec(x5, y5)
        barrier()
        dt5 = max_over_ranks(time.perf_counter() - t0) / 20
        _, mv5 = A5.footprint()
        mv5_all = sum_over_ranks(mv5)
        its5, dtc5, res25, phases5 = cg_run(A5, n5, n5g, args.c5_cg_steps, profile_phases=use_dist, tag="c5")
        moved5 = sum_over_ranks(mv5 + 64 * n5)
        c5 = {"workload": label5, "n": n5g, "nnz": int(sum_over_ranks(nnz5)), "kernel": A5.kernel,
              "spmv_ms": 1e3 * dt5, "spmv_GB/s_moved": mv5_all / dt5 / 1e9,
              "spmv_frac_of_hbm_peak": mv5_all / dt5 / 1e9 / (HBM_PEAK_GBS * world),
              "cg_iters_per_s": its5 / dtc5, "cg_iterations": its5, "cg_ms_per_iter": 1e3 * dtc5 / its5,
              "cg_GB/s_moved": moved5 * its5 / dtc5 / 1e9,
              "cg_frac_of_hbm_peak": moved5 * its5 / dtc5 / 1e9 / (HBM_PEAK_GBS * world),
              "cg_final_res2": res25, "scaling": "strong", "product_bit_exact_on_every_rank": check_c5,
              "phases": phases5, "halo_comm": bool(args.halo_comm),
              "note": "north_star target: cg_iters_per_s at n_gpus = 8 >= 6 x the n_gpus = 1 figure"}
        A5.destroy()

    # ---- N = 1: C5 as 8 in-process z-slabs on this GPU: one rank's share of a CG iteration, and the speed-up it bounds ----
    c5p = None
    if rank == 0 and world == 1 and not use_dist and not args.no_c5 and not args.no_c5_parts and args.workload == "c2" and c5 is not None:
        hb.phase(f"c5_{args.c5_parts}parts: 464^3 as {args.c5_parts} in-process parts (create, products, CG)")
        ar_ms = None
        if dist_overhead and "phases" in dist_overhead:
            ph = dist_overhead["phases"]["allreduce"]
            ar_ms = ph["ms_per_iter"] / max(1e-9, ph["events_per_iter"])
        c5p = c5_parts_leg(args, sg, torch, dev, c5, ar_ms)

    # ---- N = 1: the other single-GPU configs of BASELINE.json (C3: configs[2], C4: configs[3]) ---------------------
    c3 = c4 = None
    if rank == 0 and world == 1 and not use_dist and args.workload == "c2":
        if not args.no_c3:
            hb.phase("c3: 1-D advection-diffusion n = 1e7: SpMV, BiCGStab, GMRES(30)")
            c3 = c3_leg(sg, P, torch, dev, n=args.c3_n, iters=args.c3_iters)
        if not args.no_c4:
            hb.phase("c4: ELLPACK random digraph n = 5e6, degree 32: generate on the device, create, SpMV")
            c4 = c4_leg(sg, P, torch, dev, n=args.c4_n)

    # What the driver's record keeps of this line is the contract keys and the SCALAR entries of `roofline`, `config` and
    # `cpu_baseline` (nested objects and other top-level keys are reduced to their names): the whole metric -- SpMV GB/s AND CG
    # iterations/s on C2 -- and the other configs therefore sit in `roofline` as flat scalars (names <= 40 characters), every
    # fraction named after the bytes it is made of (see spmv_fracs); the nested legs stay in the line for readers of stdout.
    flat = flat_roofline_keys(world, cg=cg, c5=c5, c5p=c5p, ceilings=ceilings, c3=c3, c4=c4, pcg=pcg, achieved=achieved,
                              moved_rank=moved_rank, k_cold=k_cold, in_solver_ms=in_solver_ms)
    roofline_other = {"c2_cg": cg, "c3": c3, "c4": c4,
                      "c5_on_this_many_gpus": {k: v for k, v in (c5 or {}).items() if k != "phases"} or None}

    # HBM bytes per launch from the PMC counters cannot be collected inside this process; they come
    # from the committed rocprofv3 --pmc passes over this same command (profiles/<round>/), and are
    # quoted only when that summary was collected with the kernel sources that run now
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", PROFILE_TAG, "pmc_hbm_traffic.json")
    if os.path.exists(tf) and world == 1 and (args.nx, args.ny) == (3162, 3162) and args.workload == "c2":
        tj = json.load(open(tf))
        if tj.get("csrc_sha1") == kernel_sha and main_kernel.split("<")[0] in tj.get("dominant_kernel", ""):
            traffic, traffic_src = tj.get("hbm_traffic_bytes"), f"profiles/{PROFILE_TAG}/pmc_hbm_traffic.json"

    if rank == 0:
        out = {
            "metric": "SpMV GB/s (achieved HBM) + CG iters/sec on 5-pt Laplacian, N=1e7",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if args.workload == "c2" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{label} (rank 0: n={n_loc}, nnz={nnz}), fp64 SpMV y=A*x",
                       "spmv_per_step": inner, "rows_per_gpu": n_loc, "nnz_per_gpu": int(nnz),
                       "parallelism": f"row-partition x{world}",
                       "cg_dist_halo_fused": halo_mode if use_dist else None, "rccl_group_probe": group_probe,
                       "matrix_resident_bytes_per_gpu": resident_rank,
                       "reference_layout_bytes_per_gpu": 12 * nnz + 4 * (n_loc + 1),
                       "value_counts": "bytes the kernel moves by construction (stored format + x + y), not the reference layout's"},
            "effective_GBs_on_reference_bytes": effective,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, **flat,
                         "kernel": main_kernel, "moved_bytes_per_launch": moved_rank,
                         "algorithmic_bytes_per_launch_reference_layout": alg_bytes_rank,
                         "effective_GBs_on_reference_bytes": alg_bytes_rank / k_avg / 1e9,
                         "avg_launch_ms": 1e3 * k_avg, "cold_launch_ms": 1e3 * k_cold,
                         "in_solver_launch_ms": in_solver_ms,
                         "other": roofline_other,
                         "traffic_source": traffic_src, "csrc_sha1": kernel_sha,
                         "note": "frac / achieved are the WARM figure: moved_bytes_per_launch / avg_launch_ms of back-to-back "
                                 "products on the same x (the sliced kernel reads 8W+4 bytes per row of its own layout, W = 5, "
                                 "+ x once + y once).  cold_frac = the same launch after 512 MiB of unrelated writes (nothing of "
                                 "the previous product left in L2 / Infinity Cache); in_solver_frac = the launch as it runs inside "
                                 "CG.  `traffic` is 2 x FETCH_SIZE + WRITE_SIZE from the PMC passes: FETCH_SIZE counts Infinity-"
                                 "Cache hits (MI355X_MICROARCH.md), so it is fabric traffic, an upper bound of DRAM traffic"},
            "spmv_variants": variants or None, "cg": cg, "dist_overhead_1rank": dist_overhead, "c5_strong_scaling": c5,
            "c5_parts_model": c5p, "stream_ceilings": ceilings,
            "c3": c3, "c4": c4, "c1_reference_sized": c1, "pcg_time_to_solution": pcg, "cpu_baseline": cpu,
            "selfcheck": {"product_bit_exact_on_every_rank": check_main,
                          "what": "every local row of one timed-workload product == its sum evaluated with torch in stored "
                                  "order from x(i) = sin(0.001 i), on every rank (halo values included)"},
        }
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    hb.phase("teardown")
    if use_dist:
        barrier()
        comm.destroy()
        dist.destroy_process_group()
    hb.finish()
    if not check_main or (c5 is not None and not c5["product_bit_exact_on_every_rank"]) or \
            any(leg is not None and not leg["product_bit_exact"] for leg in (c3, c4)):
        sys.exit(3)          # a product that differs is not a measurement


# ------------------------------------------------------------------------------------------ #
# Two byte counts per product, kept apart by NAME everywhere in the line:
#   frac_moved        = bytes the running kernel moves by construction (sgm_mat_footprint: its stored format as it reads it
#                       + x once + y once) / time / 8 TB/s.  Never above 1.
#   frac_survey_bytes = SURVEY 8d's algorithmic bytes of the REFERENCE layout (B_csr / B_ell) / time / 8 TB/s -- reported only
#                       where the kernel moves at least those bytes (then it is a true lower bound on achieved HBM); where the
#                       kernel reads a COMPRESSED layout (moved < survey bytes) the figure would be a rate no memory system
#                       delivered, so the line carries `layout_compression` = survey / moved instead and leaves it null.
def spmv_fracs(t, moved, survey):
    out = {"spmv_ms": 1e3 * t, "moved_bytes_per_launch": int(moved), "survey_bytes_per_launch": int(survey),
           "frac_moved": moved / t / 1e9 / HBM_PEAK_GBS,
           "frac_survey_bytes": (survey / t / 1e9 / HBM_PEAK_GBS) if moved >= survey else None,
           "layout_compression": survey / moved}
    return out


def timed_launches(torch, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def fixed_iterations(sg, torch, mk, A, n, b, its):
    s = mk()
    s.set_max_iter(its)
    s.setup(A)
    u = torch.zeros(n, dtype=torch.float64, device=b.device)
    s.solve(A, u, b, check=False)
    u.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.solve(A, u, b, check=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = (int(s.last_iterations), dt, float(s.res2))
    s.destroy()
    return out


def c3_leg(sg, P, torch, dev, n=10_000_000, iters=300, gmres_orth=1):
    """BASELINE configs[2]: 1-D advection-diffusion (nonsymmetric tridiagonal, test/solver_test_advection_diffusion_1d.f90:64-82
    at n = 1e7), generated on the device: SpMV, BiCGStab and GMRES(30) for a FIXED number of iterations (SURVEY 8d C3:
    the problem does not converge at this size; iterations/s and the residual are what is reported)."""
    dx, c = 1.0 / (n + 1), 0.5
    ptr, node, val = P.tridiag_csr_torch(n, 2.0, -1.0 + c * dx / 2, -1.0 - c * dx / 2, dev)
    nnz = int(val.numel())
    torch.cuda.synchronize()
    A = sg.csr_matrix(n, n, ptr, node, val)
    x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    A.matvec(x, y)
    torch.cuda.synchronize()
    # every row against its sum in stored order (lower, diagonal, upper), torch elementwise ops: products rounded, then added
    lo, up = -1.0 - c * dx / 2, -1.0 + c * dx / 2
    z = torch.zeros(n, dtype=torch.float64, device=dev)
    z[1:] = z[1:] + lo * x[:-1]
    z = z + 2.0 * x
    z[:-1] = z[:-1] + up * x[1:]
    ok = bool(torch.equal(0.0 + z, y))
    del z, ptr, node, val
    t = timed_launches(torch, lambda: A.matvec(x, y), 50)
    _, moved = A.footprint()
    survey = spmv_bytes(n, n, nnz)
    out = {"workload": f"C3 1-D advection-diffusion CSR, n={n}, nnz={nnz} (generated on the device)", "kernel": A.kernel,
           "product_bit_exact": ok}
    out.update(spmv_fracs(t, moved, survey))
    b = torch.full((n,), 2.0 * dx * dx, dtype=torch.float64, device=dev)
    # moved bytes per iteration.  BiCGStab: two products + 12 vector passes (SURVEY 8d's fused floor is 14 passes on the
    # reference layout; the two product outputs are counted inside `moved`).  GMRES(30) with blocked CGS-2 at basis size j:
    # one product + (3 j + 6) passes; averaged over a restart cycle j = 1..30, + the cycle's x update (32 passes) and residual product
    its, dt, res2 = fixed_iterations(sg, torch, lambda: sg.bicgstab(1e-300), A, n, b, iters)
    per_it = 2 * moved + 96 * n
    out["bicgstab"] = {"iterations": its, "iters_per_s": its / dt, "ms_per_iter": 1e3 * dt / its, "final_res2": res2,
                       "moved_bytes_per_iter": per_it, "frac_moved": per_it * its / dt / 1e9 / HBM_PEAK_GBS}
    def mk_gmres():
        s = sg.gmres(1e-300, 30)
        s.set_option("gmres_cgs2", gmres_orth)
        return s
    its, dt, res2 = fixed_iterations(sg, torch, mk_gmres, A, n, b, iters)
    # vector passes at basis size j: low-synchronisation CGS-2 reads the basis twice and z twice and writes the new column
    # (2 j + 3); MGS 4 j + 8
    passes = {1: lambda j: 2 * j + 3, 0: lambda j: 4 * j + 8}[gmres_orth]
    cyc = sum(moved + 8 * n * passes(j) for j in range(1, 31)) + 8 * n * 32 + moved
    out["gmres30"] = {"iterations": its, "iters_per_s": its / dt, "ms_per_iter": 1e3 * dt / its, "final_res2": res2,
                      "moved_bytes_per_restart_cycle": cyc, "frac_moved": cyc * (its / 30.0) / dt / 1e9 / HBM_PEAK_GBS,
                      "orthogonalisation": {1: "low-synchronisation CGS-2 (basis read twice, 2 reductions per step)",
                                            0: "modified Gram-Schmidt"}[gmres_orth]}
    A.destroy()
    return out


def c4_leg(sg, P, torch, dev, n=5_000_000, d=32, both_kernels=True):
    """BASELINE configs[3]: ELLPACK random digraph, degree 32, n = 5e6 (per-row 64-bit LCG, duplicates
    rejected, val(k,i) = 1/(k + (i mod 7))), generated ON THE DEVICE (the host generator took 8.9 s): SpMV with the
    default kernel (column-blocked two-phase product) and with the plain slot-major kernel."""
    t0 = time.perf_counter()
    node, val = P.random_regular_ell_torch(n, d, 12345, dev)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    A = sg.ellpack_matrix(n, n, node, val)
    sg.synchronize()
    create_s = time.perf_counter() - t0
    x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    A.matvec(x, y)
    torch.cuda.synchronize()
    # ellpack_matvec_add's loop (ellpack_matrices.f90:652-661) with torch: all d slots in slot order, products rounded, then added
    z = torch.zeros(n, dtype=torch.float64, device=dev)
    for k in range(d):
        z = z + val[:, k] * x[(node[:, k] - 1).to(torch.int64)]
    ok = bool(torch.equal(0.0 + z, y))
    del z, node, val
    survey = 12 * n * d + 16 * n
    out = {"workload": f"C4 ELLPACK random digraph, degree {d}, n={n} (generated on the device)", "generate_s": gen_s,
           "create_s": create_s, "resident_bytes": A.footprint()[0], "product_bit_exact": ok}
    t = timed_launches(torch, lambda: A.matvec(x, y), 30)
    out["kernel"] = A.kernel
    out.update(spmv_fracs(t, A.footprint()[1], survey))
    if both_kernels:
        A.set_option("ell_colblock", 0)
        t2 = timed_launches(torch, lambda: A.matvec(x, y), 15)
        e = {"kernel": A.kernel}
        e.update(spmv_fracs(t2, A.footprint()[1], survey))
        out["slot_major_kernel"] = e
    A.destroy()
    return out


# ------------------------------------------------------------------------------------------ #
def dist_overhead_leg(args, sg, torch, dev, plain_iters_per_s):
    """CG on the workload matrix through the row-partition code path with ONE rank over the real librccl: the matrix is
    a `dist_csr_matrix` (no neighbours), every dot goes partial sums -> one-block reduce -> ncclAllReduce (forced with one
    rank: option dist_force_collectives) -> slot.  Beside the plain path's figure this is the fixed per-iteration cost the
    distributed path adds before any other GPU takes part.  Never takes the bench line down."""
    import numpy as np
    out = {"what": "CG iterations/s on the same matrix through sgm_csr_create_dist + ncclAllReduce with one rank (real librccl) "
                   "vs the plain single-GPU path"}
    try:
        comm = sg.Comm(0, 1, sg.Comm.unique_id())
        arrays = local_rows_poisson2d(args.nx, args.ny, 1, 0)
        n_loc = args.nx * args.ny
        A = sg.dist_csr_matrix(comm, np.array([0, n_loc], np.int64), *arrays)
        sg.set_option("dist_force_collectives", 1)
        try:
            s = sg.cg(1e-300)
            s.set_max_iter(args.cg_steps)
            s.setup(A)
            b = torch.full((n_loc,), 1.0 / n_loc, dtype=torch.float64, device=dev)
            u = torch.zeros(n_loc, dtype=torch.float64, device=dev)
            s.solve(A, u, b, check=False)
            u.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.solve(A, u, b, check=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            its = s.last_iterations
            u.zero_()
            sg.dist_profile(True)
            s.solve(A, u, b, check=False)
            pr = sg.dist_profile_read()
            sg.dist_profile(False)
            out.update({"iters_per_s_dist_path": its / dt, "ms_per_iter_dist_path": 1e3 * dt / its, "iterations": its,
                        "iters_per_s_plain_path": plain_iters_per_s,
                        "overhead_us_per_iter": (1e6 * dt / its - 1e6 / plain_iters_per_s) if plain_iters_per_s else None,
                        "phases": {nm: {"ms_per_iter": v["ms"] / max(1, its), "events_per_iter": v["count"] / max(1, its)}
                                   for nm, v in pr.items()}})
            s.destroy()
        finally:
            sg.set_option("dist_force_collectives", 0)
        A.destroy()
        comm.destroy()
    except Exception as e:
        out["error"] = str(e)[:300]
    return out


def stream_ceilings():
    """tools/stream_bench (built by __graft_entry__.build()) at the footprints of C2 (600 MiB) and of C5 on one GPU (7600 MiB):
    read-only, copy and the 8-reads-per-write mix of 16-byte streams an SpMV on the sliced layout amounts to -- the ceilings the
    fractions of 8 TB/s are to be read against.  A child process of its own (nothing of this process's GPU state is shared);
    None when the binary is missing.  Never takes the line down."""
    exe = os.path.join(ROOT, "tools", "stream_bench")
    if not os.path.exists(exe):
        return None
    out = {"what": "best of each kind over grids / load flavours, GB/s (r+w for copy and mix); tools/stream_bench.cpp", "flat": {}}
    try:
        for mib in (600, 7600):
            p = subprocess.run([exe, str(mib)], capture_output=True, text=True, timeout=120)
            best = {}
            for ln in p.stdout.splitlines():
                kind = "read" if ln.startswith("read") else "copy" if ln.startswith("copy") else "mix8r1w" if ln.startswith("mix 8") else \
                       "mix3r1w" if ln.startswith("mix 3") else None
                if kind:
                    best[kind] = max(best.get(kind, 0.0), float(ln.split("us")[1].split("GB/s")[0]))
            out[f"{mib}MiB"] = best
            for kind, v in best.items():
                out["flat"][f"ceiling_{kind}_{mib}MiB_GBs"] = v
    except Exception as e:
        out["error"] = str(e)[:200]
    return out


def c5_parts_leg(args, sg, torch, dev, c5, allreduce_ms):
    """The C5 matrix (7-point m^3) as P in-process z-slabs on THIS GPU (sgm_csr_create_partitioned_parts: the partition, halo
    lists, gathers, per-part kernels and reductions an 8-rank run has, all parts taking turns on one GPU): CG for the same
    fixed iteration count.  ms_per_iter / P is what ONE rank's kernels cost per iteration; with the all-reduce latency of the
    real librccl (one rank: `dist_overhead_1rank`) twice per iteration that bounds the 8-GPU speed-up from above:
        c5_model_Pgpu_speedup = t_1gpu_iter / (t_Pparts_iter / P + 2 t_allreduce)
    -- what P GPUs CAN reach before the costs only a real node shows (xGMI latency of the send / recv pairs, the all-reduce
    across 8 ranks instead of 1, load imbalance).  Never takes the line down."""
    import numpy as np
    P = args.c5_parts
    out = {"what": c5_parts_leg.__doc__.split("\n\n")[0][:200]}
    try:
        m = args.c5_edge
        zs = [(m * r) // P for r in range(P + 1)]
        starts = np.array([z * m * m for z in zs], dtype=np.int64)
        n = m ** 3
        t0 = time.perf_counter()
        parts = [local_rows_laplace3d(m, zs[k], zs[k + 1], dev) for k in range(P)]
        torch.cuda.synchronize()
        A = sg.partitioned_csr_matrix.from_parts(starts, parts)
        del parts
        torch.cuda.synchronize()
        create_s = time.perf_counter() - t0
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        A.matvec(x, y)
        # every row against the one-part product of the same x (the C5 leg checked that one against its stored-order sums)
        k = torch.arange(0, n, device=dev, dtype=torch.int64)
        pl = m * m
        i, j, l = k % m, (k // m) % m, k // pl
        z = torch.zeros(n, dtype=torch.float64, device=dev)
        for o, mask, v in [(-pl, l > 0, -1.0), (-m, j > 0, -1.0), (-1, i > 0, -1.0), (0, None, 6.0), (1, i < m - 1, -1.0),
                           (m, j < m - 1, -1.0), (pl, l < m - 1, -1.0)]:
            xo = torch.sin(0.001 * (k + (o + 1)).to(torch.float64))
            z = z + v * xo if mask is None else torch.where(mask, z + v * xo, z)
        exact = bool(torch.equal(0.0 + z, y))
        del k, i, j, l, z, xo
        t_spmv = timed_launches(torch, lambda: A.matvec(x, y), 10)
        del x, y
        b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
        res = {}
        for mode in (1, 0):
            s = sg.cg(1e-300)
            s.set_max_iter(args.c5_cg_steps)
            s.set_option("dist_halo_fused", mode)
            s.setup(A)
            u = torch.zeros(n, dtype=torch.float64, device=dev)
            s.solve(A, u, b, check=False)
            u.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.solve(A, u, b, check=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            its = s.last_iterations
            u.zero_()
            sg.dist_profile(True)
            s.solve(A, u, b, check=False)
            pr = sg.dist_profile_read()
            sg.dist_profile(False)
            res[mode] = {"ms_per_iter": 1e3 * dt / its, "iterations": its, "final_res2": s.res2,
                         "phases_ms_per_iter": {nm: v["ms"] / max(1, its) for nm, v in pr.items()},
                         "phase_events_per_iter": {nm: v["count"] / max(1, its) for nm, v in pr.items()}}
            s.destroy()
            del u
        A.destroy()
        t1 = c5["cg_ms_per_iter"]
        tp = res[1]["ms_per_iter"]
        ar = allreduce_ms if allreduce_ms is not None else 0.0
        ph = res[1]["phases_ms_per_iter"]
        out.update({
            "parts": P, "create_s": create_s, "product_bit_exact": exact,
            "cg_fused_halo": res[1], "cg_p_exchanged_by_every_product": res[0],
            "same_res2_both_modes": res[0]["final_res2"] == res[1]["final_res2"],
            f"c5_{P}parts_spmv_ms": 1e3 * t_spmv, f"c5_{P}parts_cg_ms_per_iter": tp,
            f"c5_{P}parts_cg_ms_per_iter_per_part": tp / P,
            f"c5_{P}parts_cg_ms_per_iter_mode0": res[0]["ms_per_iter"],
            f"c5_{P}parts_products_ms_per_part": (ph["interior_rows"] + ph["boundary_rows"]) / P,
            f"c5_{P}parts_halo_ms_per_part": ph["halo_post_to_done"] / P,
            f"c5_{P}parts_dot_reduce_ms_per_part": (ph["dot_reduce_kernels"] + ph["allreduce"]) / P,
            "c5_1part_cg_ms_per_iter": t1, "c5_allreduce_1rank_ms": allreduce_ms,
            f"c5_model_{P}gpu_speedup": t1 / (tp / P + 2.0 * ar),
            f"c5_model_{P}gpu_cg_iters_per_s": 1e3 / (tp / P + 2.0 * ar),
            # the same formula with an all-reduce of 30 us / 55 us (what 8 ranks over xGMI may plausibly cost; the one-rank
            # figure above is a lower bound of that term): the range a first measured SCALE line is to be read against
            f"c5_model_{P}gpu_speedup_ar30us": t1 / (tp / P + 2.0 * 0.030),
            f"c5_model_{P}gpu_speedup_ar55us": t1 / (tp / P + 2.0 * 0.055),
            "model_note": "c5_model_* are MODELLED upper bounds (in-process parts on one GPU + an assumed all-reduce latency), "
                          "not measurements of a multi-GPU run",
        })
    except Exception as e:
        out["error"] = str(e)[:300]
    return out


# ------------------------------------------------------------------------------------------ #
# What the driver's record keeps of the JSON line is the contract keys and the first 24 SCALAR entries of `roofline` (six of
# them are bound / achieved / peak / unit / frac / traffic), the scalars of `config` and of `cpu_baseline`; nested objects and
# other top-level keys are reduced to their names.  The 18 scalars that follow `traffic` are therefore chosen, in this order,
# as one line per thing BASELINE.json grades; every other figure follows and stays in the nested legs.
# tests/test_cabi_cpu.py::test_bench_flat_keys_fit_the_drivers_record holds this list.
# ------------------------------------------------------------------------------------------ #
FLAT_HEAD = (
    "cold_frac", "in_solver_frac",
    "c2_cg_iters_per_s", "c2_cg_frac_moved",
    "c3_spmv_frac_moved", "c3_bicgstab_iters_per_s", "c3_gmres30_iters_per_s",
    "c4_spmv_ms", "c4_spmv_frac_survey_bytes", "c4_product_bit_exact",
    "c5_1gpu_spmv_frac_moved", "c5_1gpu_cg_iters_per_s",
    "c5_8parts_cg_ms_per_iter_per_part", "c5_allreduce_1rank_ms", "c5_model_8gpu_speedup",
    "ceiling_copy_600MiB_GBs", "ceiling_copy_7600MiB_GBs", "pcg3162_ildu_reorder_over_cg",
)
DRIVER_KEEPS_ROOFLINE_SCALARS = 24


def flat_roofline_keys(world, cg=None, c5=None, c5p=None, ceilings=None, c3=None, c4=None, pcg=None, achieved=None,
                       moved_rank=None, k_cold=None, in_solver_ms=None):
    """Flat scalars of `roofline` (names <= 40 characters, every fraction named after the bytes it is made of): FLAT_HEAD
    first, in that order, then the rest.  With N > 1 the c5 keys carry the rank count (c5_8gpu_...) and lead the line."""
    allk = {}
    if moved_rank and k_cold:
        allk["cold_frac"] = moved_rank / k_cold / 1e9 / HBM_PEAK_GBS
    if moved_rank and in_solver_ms:
        allk["in_solver_frac"] = moved_rank / (in_solver_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if cg is not None:
        allk.update({"c2_cg_iters_per_s": cg["iters_per_s"], "c2_cg_ms_per_iter": cg["ms_per_iter"],
                     "c2_cg_frac_moved": cg["frac_of_hbm_peak"],
                     "c2_cg_eff_GBs_on_survey_floor": cg["effective_GBs_on_survey_floor"]})
    if c5 is not None:
        tag = "c5_1gpu" if world == 1 else f"c5_{world}gpu"
        allk.update({f"{tag}_spmv_ms": c5["spmv_ms"], f"{tag}_spmv_frac_moved": c5["spmv_frac_of_hbm_peak"],
                     f"{tag}_cg_iters_per_s": c5["cg_iters_per_s"], f"{tag}_cg_frac_moved": c5["cg_frac_of_hbm_peak"]})
    if c5p is not None and "error" not in c5p:
        allk.update({k: v for k, v in c5p.items() if k.startswith("c5_") and isinstance(v, (int, float))})
    if ceilings is not None:
        allk.update(ceilings["flat"])
        for key, fp in (("c2", "600"), ("c5", "7600")):
            cp = ceilings["flat"].get(f"ceiling_copy_{fp}MiB_GBs")
            mx = ceilings["flat"].get(f"ceiling_mix8r1w_{fp}MiB_GBs")
            got = achieved if key == "c2" else (c5["spmv_GB/s_moved"] if c5 else None)
            if cp and got:
                allk[f"{key}_frac_of_copy_ceiling"] = got / cp
            if mx and got:
                allk[f"{key}_frac_of_mix_ceiling"] = got / mx
    if c3 is not None:
        allk.update({"c3_spmv_ms": c3["spmv_ms"], "c3_spmv_frac_moved": c3["frac_moved"],
                     "c3_spmv_layout_compression": c3["layout_compression"],
                     "c3_bicgstab_iters_per_s": c3["bicgstab"]["iters_per_s"], "c3_bicgstab_frac_moved": c3["bicgstab"]["frac_moved"],
                     "c3_gmres30_iters_per_s": c3["gmres30"]["iters_per_s"], "c3_gmres30_frac_moved": c3["gmres30"]["frac_moved"]})
    if c4 is not None:
        allk.update({"c4_spmv_ms": c4["spmv_ms"], "c4_spmv_frac_moved": c4["frac_moved"],
                     "c4_spmv_frac_survey_bytes": c4["frac_survey_bytes"], "c4_product_bit_exact": c4["product_bit_exact"]})
    if pcg is not None:
        for key, leg in (pcg.items() if "cg" not in pcg else [("grid_1000", pcg)]):
            g = key.replace("grid_", "")
            co = leg["ildu0_colour_order"]
            allk[f"pcg{g}_cg_solve_s"] = leg["cg"]["setup_s"] + leg["cg"]["solve_s"]
            allk[f"pcg{g}_ildu_natural_total_s"] = leg["ildu0_natural_order"]["setup_s"] + leg["ildu0_natural_order"]["solve_s"]
            allk[f"pcg{g}_ildu_colour_total_s"] = co["ordering_s"] + co["permutation_s"] + co["setup_s"] + co["solve_s"]
            ro = leg["ildu0_reorder_inside_the_preconditioner"]
            allk[f"pcg{g}_ildu_reorder_total_s"] = ro["setup_s"] + ro["solve_s"]
            allk[f"pcg{g}_ildu_reorder_over_cg"] = ro["total_s_over_plain_cg_s"]
    flat = {}
    if world > 1:                          # the scaling line: what this rank count measured comes first
        for k in allk:
            if k.startswith(f"c5_{world}gpu_"):
                flat[k] = allk[k]
    for k in FLAT_HEAD:
        if k in allk:
            flat[k] = allk[k]
    for k, v in allk.items():
        flat.setdefault(k, v)
    return flat


# ------------------------------------------------------------------------------------------ #
def cpu_baseline(args, host, n_loc):
    """cpu_baseline: the REFERENCE (oracle/_ref/sigma_ref_driver = our driver linked against the
    reference's own modules, 1 thread: it has no threading) timing A%matvec on the SAME C2 matrix
    the GPU ran, + the oracle's C loop on 1 thread and on all cores (arrays first touched inside
    the OpenMP region).  About 25 s of host work in all."""
    import numpy as np
    import oracle as orc
    from sigma_amd import problems as P
    ptr, node, val = host
    Ao = orc.CsrMatrix(n_loc, n_loc, ptr, node, val)
    xv = P.test_vector(n_loc)
    alg = spmv_bytes(Ao.n, Ao.n, Ao.nnz)
    reps = 10
    sec = orc.time_csr_matvec(Ao, xv, reps)
    port = {"value": alg / sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"the full workload matrix (n={Ao.n}, nnz={Ao.nnz}), {reps} matvecs of oracle/sigma_oracle.c "
                      f"(csr_matvec_add restatement), {sec * 1e3:.1f} ms each; reference-layout bytes"}
    sec_omp, nthreads, y_omp = orc.time_csr_matvec_omp(Ao, xv, reps)
    port["all_cores_openmp"] = {"GB/s": alg / sec_omp / 1e9, "threads": nthreads, "ms_per_matvec": 1e3 * sec_omp,
                                "host_logical_cores": os.cpu_count(),
                                "note": "same row loop under one `omp parallel for schedule(static)`; private copies of "
                                        "the arrays are first touched inside the parallel region (NUMA-local)",
                                "rows_equal_single_thread": bool(np.array_equal(y_omp, Ao.matvec(xv)))}
    ref = reference_cpu_baseline(args.nx, args.ny)
    if ref:
        ref["port_on_same_matrix"] = port
        return ref
    return port


def pcg_leg(sg, P, torch, dev, nx=1000, tol=1e-8):
    """5-point grid nx^2, b = A * (a smooth vector), from u = 0 to an absolute `tol`: plain CG, ILDU(0)-PCG on the matrix as it
    is (strip-pipelined triangular sweeps) and after the reference's greedy_color_ordering + left/right permutation
    (two-level factors: row-space sweeps).  Setup (pattern, factorisation, index work: on the device) and solve timed
    separately, second solve of two."""
    import numpy as np
    n = nx * nx
    ptr, node, val = P.poisson2d_csr(nx, nx)
    xs = np.sin(np.arange(n) * 1e-3) + 1.0

    def solve(A, b, pc):
        s = sg.cg(tol)
        s.setup(A)
        u = torch.zeros(n, dtype=torch.float64, device=dev)
        s.solve(A, u, b, pc)
        torch.cuda.synchronize()
        u.zero_()
        t0 = time.perf_counter()
        s.solve(A, u, b, pc)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, int(s.last_iterations), u

    out = {"workload": f"5-point {nx}x{nx} grid (n={n}), CG from 0 to an absolute {tol:g}; seconds"}
    A = sg.csr_matrix(n, n, ptr, node, val)
    bh = np.zeros(n)
    A.matvec(xs, bh)
    b = torch.from_numpy(bh).to(dev)
    t, it, u = solve(A, b, None)
    out["cg"] = {"setup_s": 0.0, "solve_s": t, "iterations": it}
    err_cg = float(np.abs(u.cpu().numpy() - xs).max())
    t0 = time.perf_counter()
    pc = sg.ldu()
    pc.setup(A)
    sg.synchronize()
    ts = time.perf_counter() - t0
    t, it, u = solve(A, b, pc)
    out["ildu0_natural_order"] = {"setup_s": ts, "solve_s": t, "iterations": it, "triangular_sweeps": "strip pipeline" if pc.get("strips", np.int32)[0] else "level walkers"}
    pc.destroy()
    t0 = time.perf_counter()
    p, _ptrs, nc = A.greedy_color_ordering()
    t1 = time.perf_counter()
    A.left_permute(p)
    A.right_permute(p)
    sg.synchronize()
    t2 = time.perf_counter()
    bp = np.empty(n)
    bp[p - 1] = bh
    b = torch.from_numpy(bp).to(dev)
    t0s = time.perf_counter()
    pc = sg.ldu()
    pc.setup(A)
    sg.synchronize()
    ts = time.perf_counter() - t0s
    t, it, u = solve(A, b, pc)
    xp = np.empty(n)
    xp[p - 1] = xs
    out["ildu0_colour_order"] = {"ordering_s": t1 - t0, "permutation_s": t2 - t1, "colours": int(nc), "setup_s": ts, "solve_s": t, "iterations": it,
                                 "row_space_levels": [int(v) for v in pc.get("row_levels", np.int32)],
                                 "max_err_vs_the_vector_b_was_made_from": float(np.abs(u.cpu().numpy() - xp).max())}
    out["cg"]["max_err_vs_the_vector_b_was_made_from"] = err_cg
    pc.destroy()
    A.destroy()
    # the same colour-ordered factorisation INSIDE the preconditioner (option ildu_reorder / sg.ldu(reorder="colour")): the matrix,
    # b and u stay in natural order -- nothing for the caller to permute; ordering (on the device for this bipartite grid),
    # permuted scratch copy and factorisation are all part of setup_s
    A = sg.csr_matrix(n, n, ptr, node, val)
    b = torch.from_numpy(bh).to(dev)
    pw = sg.ldu(reorder="colour"); pw.setup(A); pw.destroy()        # (code objects loaded once, like every other leg's warm-up solve)
    sg.synchronize()
    t0 = time.perf_counter()
    pc = sg.ldu(reorder="colour")
    pc.setup(A)
    sg.synchronize()
    ts = time.perf_counter() - t0
    t, it, u = solve(A, b, pc)
    rm = pc.get("reorder_ms", np.float64)
    out["ildu0_reorder_inside_the_preconditioner"] = {
        "setup_s": ts, "of_which_ordering_s": rm[0] * 1e-3, "of_which_permuted_copy_s": rm[1] * 1e-3, "colours": int(rm[3]),
        "solve_s": t, "iterations": it, "max_err_vs_the_vector_b_was_made_from": float(np.abs(u.cpu().numpy() - xs).max()),
        "total_s_over_plain_cg_s": (ts + t) / max(1e-12, out["cg"]["setup_s"] + out["cg"]["solve_s"])}
    pc.destroy()
    A.destroy()
    return out


def c1_leg(sg, P, torch, dev, n=10000, tol=1e-16):
    """BASELINE configs[0]: tridiag(-1, 2, -1), n = 10,000, f = 2 dx^2, CG from u = 0 to an absolute 1e-16 -- the matrix
    assembled in the reference's insertion order; the device solve (second of two) beside the reference itself solving
    the same system on one host core (oracle/_ref/sigma_ref_driver), both against the analytic solution."""
    import struct
    import tempfile
    import numpy as np
    (ei, ej, ev), f, v = P.diffusion_1d(n)
    A = sg.csr_matrix.from_edges(n, n, ei, ej, ev)          # device-side assembly in the reference's insertion order
    b = torch.from_numpy(f).to(dev)
    s = sg.cg(tol)
    s.setup(A)
    u = torch.zeros(n, dtype=torch.float64, device=dev)
    s.solve(A, u, b)
    torch.cuda.synchronize()
    u.zero_()
    t0 = time.perf_counter()
    s.solve(A, u, b)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    out = {"workload": f"tridiag(-1,2,-1) n={n}, CG from 0 to an absolute {tol:g} (BASELINE configs[0])",
           "gpu_ms": 1e3 * gpu_s, "gpu_iterations": int(s.last_iterations),
           "gpu_max_err_vs_analytic": float(np.abs(u.cpu().numpy() - v).max())}
    # the same solve with the reference's dot_product order (option dot_order = 1: one accumulator, first element to last):
    # the iterates are then the reference's bit for bit -- it stops where the reference stops
    s.set_option("dot_order", 1)
    u.zero_()
    s.solve(A, u, b)
    torch.cuda.synchronize()
    u.zero_()
    t0 = time.perf_counter()
    s.solve(A, u, b)
    torch.cuda.synchronize()
    out["gpu_dot_order1_ms"] = 1e3 * (time.perf_counter() - t0)
    out["gpu_dot_order1_iterations"] = int(s.last_iterations)
    out["gpu_dot_order1_max_err_vs_analytic"] = float(np.abs(u.cpu().numpy() - v).max())
    s.destroy()
    A.destroy()
    drv = os.path.join(ROOT, "oracle", "_ref", "sigma_ref_driver")
    if os.path.exists(drv):
        try:
            with tempfile.TemporaryDirectory() as td:
                inp = os.path.join(td, "in.bin")
                with open(inp, "wb") as fh:
                    fh.write(struct.pack("<5i", n, n, len(ei), 1, 1))
                    fh.write(np.asarray(ei, "<i4").tobytes())
                    fh.write(np.asarray(ej, "<i4").tobytes())
                    fh.write(np.asarray(ev, "<f8").tobytes())
                    fh.write(np.zeros(n, "<f8").tobytes())
                    fh.write(np.asarray(f, "<f8").tobytes())
                    fh.write(struct.pack("<iid", 1, 0, tol))
                o = subprocess.run([drv, inp, os.path.join(td, "o"), "time:1"], capture_output=True, text=True, timeout=300)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith("solve 1:")][0]
            out["reference_cpu_ms"] = 1e3 * float(line.split("seconds=")[1].split()[0])
            out["reference_iterations"] = int(line.split("iterations=")[1].split()[0])
            out["reference_kind"] = "danshapero/sigma itself (amdflang -O2), one host core"
        except Exception as e:        # a baseline leg must never take the bench line down
            sys.stderr.write(f"[bench] C1 reference leg skipped: {e}\n")
    return out


def reference_cpu_baseline(nx, ny, reps=10, nx_cg=600):
    drv = os.path.join(ROOT, "oracle", "_ref", "sigma_ref_driver")
    if not os.path.exists(drv):
        return None
    import struct
    import tempfile
    import numpy as np
    from sigma_amd import problems as P
    try:
        t0 = time.time()
        out = subprocess.run([drv, f"gen:poisson2d:{nx}:{ny}", "-", f"time:{reps}"], capture_output=True, text=True, timeout=600)
        wall = time.time() - t0
        sec = float(out.stdout.split("matvec_seconds_each=")[1].split()[0])
        n = nx * ny
        nnz = 5 * n - 2 * nx - 2 * ny
        # one unpreconditioned CG solve to 1e-8 on a smaller grid (the reference has no iteration cap)
        n2 = nx_cg * nx_cg
        ei, ej, ev = P.poisson2d_edges(nx_cg, nx_cg)
        with tempfile.TemporaryDirectory() as td:
            inp = os.path.join(td, "in.bin")
            with open(inp, "wb") as f:
                f.write(struct.pack("<5i", n2, n2, len(ei), 1, 1))
                f.write(np.asarray(ei, "<i4").tobytes())
                f.write(np.asarray(ej, "<i4").tobytes())
                f.write(np.asarray(ev, "<f8").tobytes())
                f.write(np.asarray(P.test_vector(n2), "<f8").tobytes())
                f.write(np.full(n2, 1.0 / n2, "<f8").tobytes())
                f.write(struct.pack("<iid", 1, 0, 1e-8))
            out2 = subprocess.run([drv, inp, os.path.join(td, "o"), "time:1"], capture_output=True, text=True, timeout=300)
        line = [ln for ln in out2.stdout.splitlines() if ln.startswith("solve 1:")][0]
        its = int(line.split("iterations=")[1].split()[0])
        cg_sec = float(line.split("seconds=")[1].split()[0])
    except Exception as e:        # a baseline leg must never take the bench line down
        sys.stderr.write(f"[bench] reference baseline skipped: {e}\n")
        return None
    return {"value": spmv_bytes(n, n, nnz) / sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
            "sample": f"danshapero/sigma itself (amdflang -O2, oracle/build_ref.sh), 1 thread: csr A%matvec on the SAME "
                      f"{nx}x{ny} 5-point matrix the GPU ran (n={n}, nnz={nnz}; assembled by the reference through "
                      f"ll_graph%add_edge / set_value), {reps} calls, {sec * 1e3:.1f} ms each ({wall:.0f} s incl. assembly); "
                      f"cg%solve on {nx_cg}^2 to 1e-8: {its} iterations in {cg_sec:.2f} s",
            "ms_per_matvec": 1e3 * sec, "cg_iters_per_s": its / cg_sec if cg_sec > 0 else None, "cg_n": n2}


def local_rows_laplace3d(m, z0, z1, dev):
    """Rows of the planes [z0, z1) of the m^3 7-point grid, built on the device: local 1-based
    ptr, GLOBAL 1-based node, val; entry order -z,-y,-x,C,+x,+y,+z like problems.laplace3d_csr."""
    import torch
    pl = m * m
    k = torch.arange(z0 * pl, z1 * pl, device=dev, dtype=torch.int64)
    i, j, l = k % m, (k // m) % m, k // pl
    one = torch.ones_like(k, dtype=torch.bool)
    offs = [(-pl, l > 0, -1.0), (-m, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 6.0), (1, i < m - 1, -1.0),
            (m, j < m - 1, -1.0), (pl, l < m - 1, -1.0)]
    mask = torch.stack([mk for _, mk, _ in offs], dim=1)
    ptr = torch.ones(k.numel() + 1, dtype=torch.int64, device=dev)
    ptr[1:] += torch.cumsum(mask.sum(dim=1), 0)
    cols = torch.stack([k + 1 + o for o, _, _ in offs], dim=1)[mask].to(torch.int32)
    vals = torch.tensor([v for _, _, v in offs], dtype=torch.float64, device=dev).expand(k.numel(), -1)[mask].contiguous()
    return ptr.to(torch.int32), cols, vals


def local_rows_poisson2d(nx, ny, world, rank):
    """Rows [rank*nx*ny, (rank+1)*nx*ny) of the nx x (world*ny) 5-point grid: local 1-based
    ptr, GLOBAL 1-based node, val -- same insertion order S,W,C,E,N as problems.poisson2d_csr."""
    import numpy as np
    n_loc = nx * ny
    k = np.arange(rank * n_loc, (rank + 1) * n_loc, dtype=np.int64)
    i, j = k % nx, k // nx
    NY = ny * world
    offs = [(-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, np.ones(n_loc, bool), 4.0), (1, i < nx - 1, -1.0),
            (nx, j < NY - 1, -1.0)]
    cols = np.stack([k + 1 + o for o, _, _ in offs], axis=1)
    mask = np.stack([m for _, m, _ in offs], axis=1)
    vals = np.broadcast_to(np.array([v for _, _, v in offs]), cols.shape)
    ptr = np.concatenate([[1], 1 + np.cumsum(mask.sum(axis=1))]).astype(np.int32)
    return ptr, cols[mask].astype(np.int32), vals[mask].astype(np.float64)


if __name__ == "__main__":
    main()
