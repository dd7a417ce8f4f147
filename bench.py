#!/usr/bin/env python3
"""Headline benchmark: reads/sec of the signal-to-reference alignment path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 launched through
torch.distributed.run with one rank per GPU.  A *step* is one pass of the hot path
(``refine_alignment``, transitions on — the ``align_signal`` path) over one batch of synthetic
reads that is already resident in HBM.  At N=1 the batch is BASELINE.json configs[1]:
10 000 reads, ~4 000 samples each, bandwidth 150, packaged 6-mer model.  Reads shard
embarrassingly: every rank aligns its own batch (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline":     HBM roofline of the banded-DP kernel — algorithmic bytes (SURVEY.md §8d,
                  B_align from the run's actual bands) / its HIP-event-timed launch duration
  "cpu_baseline": the CPU oracle (oracle/, the reference compiled in place when oracle/_ref is
                  present, else the C restatement) timed on this box's host cores on a bounded
                  sample of the same reads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(batch, model, bandwidth, mel, workload, budget_reads_per_core=256):
    """Time the CPU oracle on a bounded sample with one thread per host core (ctypes releases
    the GIL; the reference itself is single-threaded)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle, have_reference
    kind = 'reference' if have_reference() else 'port'
    o = Oracle(kind)
    m = o.KmerModel(*model)
    # a 1-GPU box's CPU share is 16 cores, whatever os.cpu_count() says; NADAVCA_CPU_THREADS overrides
    cores = max(1, min(os.cpu_count() or 1, 16))
    if hasattr(os, 'sched_getaffinity'):
        cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    cores = int(os.environ.get('NADAVCA_CPU_THREADS', cores))
    per_core = budget_reads_per_core if workload == 'cfg2_align' else (
        1 if workload == 'cfg5_long' else max(8, budget_reads_per_core // 10))
    n = min(batch.n, cores * per_core)
    cases = batch.cases[:n]

    def work(c):
        a = (c['signal'], c['reference'], c['context_before'], c['context_after'],
             c['approximate_alignment'], bandwidth, mel, m)
        if workload in ('cfg2_align', 'cfg5_long'):
            o.refine_alignment(*a, True)
        else:
            o.estimate_log_likelihoods(*a, True)

    work(cases[0])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, cases))
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'reads/s', 'cores': cores, 'kind': kind,
            'sample': '%d of the run\'s reads, %s, %d threads, %.1f s wall' % (
                n, 'estimate_log_likelihoods(wobbling)' if workload == 'cfg3_snps' else 'refine_alignment(transitions)',
                cores, dt),
            'per_core': n / dt / cores}


def measured_traffic(workload, n_reads, key='bytes_per_launch'):
    """HBM bytes per launch of the dominant kernel (or another recorded figure, `key`) from the
    committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json; counters cannot be read from inside
    this process).  None when no measurement exists for this workload/size."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.json')), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get('workload') == workload and int(t.get('reads_per_launch', -1)) == int(n_reads):
            return t.get(key)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='cfg2_align', choices=['cfg2_align', 'cfg3_snps', 'cfg5_long'],
                    help='cfg2_align (default, the headline), cfg3_snps, cfg5_long (BASELINE config 5 shape: '
                         '~50k-sample reads, bandwidth 1000; refine_alignment)')
    ap.add_argument('--reads', type=int, default=0, help='reads per GPU per step (default: the config size)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--slots', type=int, default=0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    from nadavca_amd import dtw, synthetic, _lib
    from nadavca_amd.device import DeviceBatch, refine_alignment_dev, estimate_log_likelihoods_dev

    wl = dict(synthetic.WORKLOADS[args.workload])
    n_reads = args.reads or wl.pop('n_reads')
    wl.pop('n_reads', None)
    bandwidth, mel = wl['bandwidth'], 2
    model = synthetic.load_model_arrays()
    ctx = _lib.Context(local_rank)
    if args.slots:
        ctx.set_slots(args.slots)
    km = dtw.KmerModel(*model, context=ctx)
    # every rank gets its own reads (seed offset by rank): weak scaling over independent reads
    batch = synthetic.make_batch(n_reads, model, seed=1000 + rank, **wl)
    dbatch = DeviceBatch(batch, device)
    events = torch.zeros((dbatch.total_ref, 2), dtype=torch.int32, device=device)
    is_align = args.workload in ('cfg2_align', 'cfg5_long')
    ll = None if is_align else torch.zeros((dbatch.total_ref, 4), dtype=torch.float64, device=device)
    status = torch.zeros(dbatch.n, dtype=torch.int32, device=device)

    def step():
        if is_align:
            refine_alignment_dev(dbatch, bandwidth, mel, km, True, events, status)
        else:
            estimate_log_likelihoods_dev(dbatch, bandwidth, mel, km, True, ll, status)

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    ctx.timing_enable(True)
    ctx.timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.timing_enable(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    n_ok = int((status == 0).sum().item())
    timing = ctx.timing_read()
    stats = ctx.last_batch_stats()
    if rank == 0:
        total_reads = n_reads * world * args.steps
        out = {
            'metric': {'cfg2_align': 'reads/sec (align_signal, ~4k-sample reads)',
                       'cfg3_snps': 'reads/sec (estimate_snps log-likelihoods, ~4k-sample reads)',
                       'cfg5_long': 'reads/sec (align_signal, ~50k-sample reads, wide band)'}[args.workload],
            'value': total_reads / dt, 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'reads_per_gpu_per_step': n_reads,
                       'samples_per_read': round(dbatch.total_signal / n_reads, 1),
                       'bases_per_read': round(dbatch.total_ref / n_reads, 1), 'bandwidth': bandwidth,
                       'min_event_length': mel, 'kmer_model': 'packaged 6-mer', 'reads_ok': n_ok,
                       'band_cells_per_read': round(stats['band_cells'] / n_reads, 1),
                       'reads_redone_exact': stats['reads_redone_exact']},
        }
        kname = 'align' if is_align else 'ell_hyp'
        ms, launches = timing[kname]
        if launches:
            algo = (dbatch.algorithmic_bytes_align(stats['band_cells']) if is_align
                    else dbatch.algorithmic_bytes_snp(stats['band_cells']))
            sec = ms / 1000.0 / launches
            ach = algo / sec / 1e9
            out['roofline'] = {'bound': 'hbm', 'kernel': kname, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                               'traffic': measured_traffic(args.workload, n_reads),
                               # what actually limits the kernel: share of cycles its SIMDs issue
                               # vector instructions (same PMC passes)
                               'valu_busy': measured_traffic(args.workload, n_reads, 'valu_busy_frac'),
                               'algorithmic_bytes_per_launch': algo, 'kernel_ms_per_launch': ms / launches,
                               'all_kernels_ms': {k: v[0] / max(v[1], 1) for k, v in timing.items() if v[1]}}
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (contract)
            out['cpu_baseline'] = cpu_baseline(batch, model, bandwidth, mel, args.workload)
            out['gpu_over_cpu'] = out['value'] / world / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
