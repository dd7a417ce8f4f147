#!/usr/bin/env python3
"""Headline benchmark: reads/sec of the signal-to-reference alignment path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 launched through
torch.distributed.run with one rank per GPU.  A *step* is one pass of the hot path over one batch of
synthetic reads that is already resident in HBM.  Workloads (``--workload``):

  cfg2_align (default)  BASELINE.json configs[1]: ``refine_alignment`` (transitions on — the ``align_signal``
                        path) over 10 000 reads, ~4 300 samples each, bandwidth 150, packaged 6-mer model.
                        Reads shard embarrassingly: every rank aligns its own batch (weak scaling, no
                        data-path collective).
  cfg3_snps             configs[2]: ``estimate_log_likelihoods`` (wobbling on) over the same shape.
  cfg4_consensus        configs[3]: the ``estimate_snps(independent=False)`` data path, 200 000 reads over the ranks
                        (strong scaling: 200 000 / N each) — pooled median / MAD over ALL ranks' samples (exact
                        distributed selection, <= 32 tiny all-reduces) -> normalise -> log-likelihoods -> strand-flip /
                        scatter-add into the per-position sums (device) -> ONE reduce(sum) of the packed [L, 5] f64
                        buffer (RCCL over xGMI when N > 1) -> posterior on the root.
  cfg5_long             configs[4] shape on one GPU: ~50 000-sample reads, bandwidth 1000.
  api_align_signal      the public ``nadavca_amd.align_signal()`` call itself (host objects in, host arrays
                        out: normalisation, two alignments, two linear re-fits) over a ``ReadBatch``.

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline":     the dominant kernel against the roofline that bounds it — align: HBM, algorithmic bytes
                  (B_align from the run's actual bands) / its HIP-event-timed duration (the
                  reverse-sweep and forward-sweep launches together); SNP kernels: FP64 vector issue
  "e2e":          (cfg2_align, N=1) host arrays in -> host arrays out: a stream of batches through
                  nvk_refine_alignment_submit/_wait (PCIe copies behind the kernels), and "e2e_single_call" for one
                  nvk_refine_alignment_batch call; never `value`
  "cpu_baseline": the CPU oracle (oracle/_ref = the reference compiled in place when present, else the C
                  restatement) timed on this box's host cores on a bounded sample of the same reads.
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
# vector-instruction issue peak at the 2.4 GHz maximum clock (MI355X_MICROARCH.md), 1024 SIMDs x 64 lanes:
# Issue bound of the log-likelihood kernel's instruction mix.  A wave64 FP64 instruction (and the 32-bit compares,
# selects, max / shift-add forms) occupies its SIMD for 4 cycles, the plain 32-bit integer ones (v_add_u32,
# v_sub_u32, v_and_b32, v_mov_b32 incl. its DPP form) for 2 (profiles/ubench_valu_gfx950.txt: 1.06-1.14 ns against
# 1.85-2.3 ns per wave-instruction).  The hypothesis loop, 5/6 of the kernel's instructions, holds 266 of the
# second kind among its 854 per 12-step trip (hipcc -S listing of kernels_ell.hip, tools/isa_blocks.py): 3.38
# cycles per instruction on average.  (Rounds 1-2 used 4 cycles for every instruction; the kernel now runs close
# enough to the bound for the difference to show: that figure would read above 1 on 200 000 reads.)
VALU_CYCLES_PER_INST = (588 * 4 + 266 * 2) / 854.0
VALU_PEAK_TLANE = 1024 * 64 * 2.4e9 / VALU_CYCLES_PER_INST / 1e12


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
    align = workload in ('cfg2_align', 'cfg5_long', 'api_align_signal')
    per_core = budget_reads_per_core if workload in ('cfg2_align', 'api_align_signal') else (
        1 if workload == 'cfg5_long' else max(8, budget_reads_per_core // 10))
    n = min(batch.n, cores * per_core)
    cases = batch.cases[:n]

    def work(c):
        a = (c['signal'], c['reference'], c['context_before'], c['context_after'],
             c['approximate_alignment'], bandwidth, mel, m)
        if align:
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
                n, 'refine_alignment(transitions)' if align else 'estimate_log_likelihoods(wobbling)',
                cores, dt),
            'per_core': n / dt / cores}


def kernel_source_hash(root):
    """sha1 over the HIP sources and headers of the library with comments and white space removed (the code the
    counters describe; editing a comment does not make a profile stale)."""
    import glob, hashlib, re
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(root, 'nadavca_amd', 'csrc', '*.hip')) +
                       glob.glob(os.path.join(root, 'nadavca_amd', 'csrc', '*.h')) +
                       glob.glob(os.path.join(root, 'include', '*.h'))):
        text = open(path, 'r', encoding='utf-8', errors='replace').read()
        text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
        text = re.sub(r'//[^\n]*', ' ', text)
        text = re.sub(r'\s+', ' ', text)
        h.update(os.path.basename(path).encode())
        h.update(text.encode())
    return h.hexdigest()


PROFILE_FIGURES_OFF = False  # set for runs the committed counter profiles do not describe (--k: other kernels)


def profile_figure(workload, key):
    """A figure that cannot be read from inside this process (HBM bytes, instruction counts: PMC counters)
    from the newest committed rocprofv3 summary profiles/*_counters.json for this workload, per read, with
    its provenance: -> (value per read or None, source file or None)."""
    import glob
    if PROFILE_FIGURES_OFF:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_counters.json')), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get('workload') == workload and key in t:
            # figures of another build of the kernels are not quoted (no stale canned numbers)
            if t.get('kernel_source_sha1') not in (None, kernel_source_hash(ROOT)):
                return None, os.path.relpath(path, ROOT) + ' (stale: other kernel sources, not quoted)'
            return t[key], os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='cfg2_align',
                    choices=['cfg2_align', 'cfg3_snps', 'cfg4_consensus', 'cfg5_long', 'api_align_signal',
                             'api_estimate_snps'])
    ap.add_argument('--no-tweak', action='store_true', help='api_estimate_snps: tweak_signal_normalization off')
    ap.add_argument('--spline-fit', choices=('device', 'host'), default='device',
                    help='api_estimate_snps: the spline fits on the device (nvk_spline_fit_dev) or by scipy on the host')
    ap.add_argument('--fit-workers', type=int, default=16,
                    help='api_estimate_snps --spline-fit host: processes for the spline fits')
    ap.add_argument('--reads', type=int, default=0, help='reads per GPU per step (default: the config size)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the host-pointer legs (profiling runs: their launches '
                    'would be averaged into the kernel statistics)')
    ap.add_argument('--slots', type=int, default=0)
    ap.add_argument('--k', type=int, default=0, help='use a synthetic k-mer table of this size (e.g. 10: the size of the '
                    "reference's coded default table, 4^10 rows) instead of the packaged 6-mer table")
    args = ap.parse_args()
    global PROFILE_FIGURES_OFF
    PROFILE_FIGURES_OFF = bool(args.k)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:   # under torch.distributed.run: also with one rank
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
    from nadavca_amd.device import (DeviceBatch, refine_alignment_dev, estimate_log_likelihoods_dev,
                                    consensus_accumulate_dev, posterior_segments_dev)

    wname = args.workload
    wl = dict(synthetic.WORKLOADS['cfg2_align' if wname.startswith('api_') else wname])
    n_reads = args.reads or wl.pop('n_reads')
    wl.pop('n_reads', None)
    if wname == 'cfg4_consensus' and not args.reads:
        n_reads = max(1, n_reads // world)   # BASELINE config 4: 200 000 reads sharded over the ranks
    ref_len = wl.pop('reference_length', 10000)
    bandwidth, mel = wl['bandwidth'], 2
    model = synthetic.synth_model_arrays(7, k=args.k, central=(args.k - 1) // 2) if args.k else synthetic.load_model_arrays()
    model_name = 'synthetic %d-mer' % args.k if args.k else 'packaged 6-mer'
    ctx = _lib.Context(local_rank)
    if args.slots:
        ctx.set_slots(args.slots)
    km = dtw.KmerModel(*model, context=ctx)
    is_align = wname in ('cfg2_align', 'cfg5_long')
    is_api = wname.startswith('api_')
    extra = {}

    if is_api:
        # the public call over a struct-of-arrays batch of simulated reads (nadavca_amd/readbatch.py)
        from nadavca_amd.align_signal import align_signal_batch
        rb, aligner, genome = synthetic.make_read_batch(n_reads, model, seed=1000 + rank, genome_length=ref_len)
        batch = None

        if wname == 'api_align_signal':
            def step():
                out = align_signal_batch(None, rb, kmer_model=km, aligner=aligner)
                extra['reads_ok'] = int(out.n_aligned)
        else:
            from nadavca_amd.estimate_snps import estimate_snps_batch
            from nadavca_amd.align_signal import _load_config
            from nadavca_amd import defaults
            cfg_snps = dict(_load_config(defaults.CONFIG_FILE), tweak_signal_normalization=not args.no_tweak)

            def step():
                chunks = estimate_snps_batch(genome, rb, config=cfg_snps, kmer_model=km, aligner=aligner,
                                             fit_workers=args.fit_workers, spline_fit=args.spline_fit)
                from nadavca_amd.estimate_snps import last_batch_counts
                extra['reads_ok'] = last_batch_counts.get('reads_ok')       # status 0 after the log-likelihoods
                extra['reads_fitted'] = last_batch_counts.get('reads_fitted')
                extra['chunks'] = len(chunks)
        stats_of = lambda: ctx.last_batch_stats()
    else:
        # every rank gets its own reads (seed offset by rank): weak scaling over independent reads
        # (config 4: at most 10 000 simulated reads per rank, repeated to the rank's share — synthetic.tile_batch)
        batch = synthetic.make_batch(min(n_reads, 10000), model, seed=1000 + rank, **wl)
        if n_reads > batch.n:
            batch = synthetic.tile_batch(batch, n_reads)
        dbatch = DeviceBatch(batch, device)
        events = torch.zeros((dbatch.total_ref, 2), dtype=torch.int32, device=device)
        ll = None if is_align else torch.zeros((dbatch.total_ref, 4), dtype=torch.float64, device=device)
        status = torch.zeros(dbatch.n, dtype=torch.int32, device=device)
        if wname == 'cfg4_consensus':
            # where each read's chunk lies on the 10 kb reference, its strand; the groups of overlapping
            # chunks over ALL ranks' reads (fixed by the intervals: set up once, estimator.py:205-220)
            from nadavca_amd.estimator import ProbabilityEstimator
            from nadavca_amd import distributed as D
            rng = np.random.default_rng(5000 + rank)
            R = np.diff(batch.ref_off)
            start = rng.integers(0, np.maximum(ref_len - R, 1))
            rev = (np.arange(n_reads) % 2).astype(np.int32)
            ranges = [(int(s), int(min(ref_len, s + r))) for s, r in zip(start, R)]
            all_ranges = D.gather_ranges(ranges, device=device) if dist is not None else ranges
            groups = ProbabilityEstimator.group_ranges(all_ranges)
            seg = np.zeros(len(groups) + 1, dtype=np.int64)
            np.cumsum([e - s for s, e in groups], out=seg[1:])
            pos = np.concatenate([np.arange(s, e) for s, e in groups]) if groups else np.zeros(0, dtype=np.int64)
            d_start = torch.from_numpy(start.astype(np.int64)).to(device)
            d_rev = torch.from_numpy(rev).to(device)
            d_pos = torch.from_numpy(pos).to(device)
            d_seg = torch.from_numpy(seg).to(device)
            d_refnum = torch.from_numpy(np.random.default_rng(4).integers(0, 4, ref_len).astype(np.int32)).to(device)[d_pos]
            acc = torch.zeros((ref_len, 4), dtype=torch.float64, device=device)
            cov = torch.zeros(ref_len, dtype=torch.int64, device=device)
            extra['groups'] = len(groups)
            # the reads as a sequencer hands them over (raw = 12 x + 90, make_read_spec's scale): every step starts
            # with the pooled median / MAD over ALL ranks' samples (estimate_snps.py:61) — the path's first exchange
            from nadavca_amd.device import select_hist_dev, normalize_apply_dev
            d_raw = dbatch.signal * 12.0 + 90.0

        def step():
            if is_align:
                refine_alignment_dev(dbatch, bandwidth, mel, km, True, events, status)
            elif wname == 'cfg3_snps':
                estimate_log_likelihoods_dev(dbatch, bandwidth, mel, km, True, ll, status)
            else:
                cs = D.pooled_centre_scale(select_hist_dev(ctx, d_raw), d_raw.numel(), device=device)
                normalize_apply_dev(ctx, d_raw, cs[0], cs[1], out=dbatch.signal)
                extra['centre_scale'] = cs
                estimate_log_likelihoods_dev(dbatch, bandwidth, mel, km, True, ll, status)
                acc.zero_()
                cov.zero_()
                consensus_accumulate_dev(ctx, dbatch, ll, d_start, d_rev, status, 10.0, ref_len, acc, cov)
                tot = (acc, cov) if dist is None else D.reduce_consensus_tensors(acc, cov, dst=0)
                if tot is not None and d_pos.numel():
                    extra['posterior'] = posterior_segments_dev(ctx, tot[0][d_pos], d_refnum, d_seg, model[0], 0.001)
        stats_of = lambda: ctx.last_batch_stats()

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

    n_ok = extra.get('reads_ok') if is_api else int((status == 0).sum().item())
    timing = ctx.timing_read()
    stats = stats_of()
    if rank == 0:
        total_reads = n_reads * world * args.steps
        metric = {'cfg2_align': 'reads/sec (align_signal, ~4k-sample reads)',
                  'cfg3_snps': 'reads/sec (estimate_snps log-likelihoods, ~4k-sample reads)',
                  'cfg4_consensus': 'reads/sec (estimate_snps independent=False: log-likelihoods + consensus '
                                    'reduce + posterior, ~4k-sample reads)',
                  'cfg5_long': 'reads/sec (align_signal, ~50k-sample reads, wide band)',
                  'api_align_signal': 'reads/sec (nadavca_amd.align_signal() end to end, ~4k-sample reads)',
                  'api_estimate_snps': 'reads/sec (nadavca_amd.estimate_snps(independent=False) end to end, '
                                       '~4k-sample reads)'}[wname]
        cfg = {'workload': wname, 'reads_per_gpu_per_step': n_reads, 'bandwidth': bandwidth,
               'min_event_length': mel, 'kmer_model': model_name, 'reads_ok': n_ok}
        if not is_api:
            cfg.update({'samples_per_read': round(dbatch.total_signal / n_reads, 1),
                        'bases_per_read': round(dbatch.total_ref / n_reads, 1),
                        'band_cells_per_read': round(stats['band_cells'] / n_reads, 1),
                        'wave_steps_per_read': round(stats['wave_steps'] / n_reads, 1)})
        if wname == 'api_estimate_snps':
            cfg.update({'tweak_signal_normalization': not args.no_tweak, 'spline_fit': args.spline_fit,
                        'fit_workers': args.fit_workers if args.spline_fit == 'host' else None,
                        'chunk_groups': extra.get('chunks'), 'reads_spline_fitted': extra.get('reads_fitted')})
        if is_align or wname == 'api_align_signal':
            cfg.update({'reads_redone_exact': stats['reads_redone_exact'],
                        # reads in which a path comparison fell inside the tie margin (include/nadavca_hip.h):
                        # exactly equal scores / within 64 ulps of the reference's log value / further apart but
                        # inside 2^-24 relative
                        'reads_tie_exact': stats['reads_tie_exact'], 'reads_tie_ulp': stats['reads_tie_ulp'],
                        'reads_tie_near': stats['reads_tie_near']})
        if wname == 'cfg4_consensus':
            cfg.update({'reference_length': ref_len, 'chunk_groups': extra.get('groups'),
                        'reads_total': n_reads * world, 'unique_simulated_reads_per_rank': min(n_reads, 10000),
                        'pooled_centre_scale': list(extra.get('centre_scale', ())),
                        'collectives': 'none (1 rank)' if dist is None else
                                       'pooled median/MAD: <= 32 all-reduces of 256 counts; consensus: ONE '
                                       'reduce(sum) of [L,5] f64; both over RCCL'})
        out = {'metric': metric, 'value': total_reads / dt, 'unit': 'reads/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps,
               'higher_is_better': True, 'scaling': 'strong' if wname == 'cfg4_consensus' and not args.reads else 'weak',
               'vs_baseline': None, 'dtype': 'f64',
               'data': 'synthetic', 'config': cfg}
        kname = 'align' if (is_align or wname == 'api_align_signal') else 'ell_hyp'
        ms, launches = timing[kname]
        per_kernel = {k: v[0] / max(v[1], 1) for k, v in timing.items() if v[1]}
        if launches and is_align:
            algo = dbatch.algorithmic_bytes_align(stats['band_cells'])
            sec = ms / 1000.0 / launches
            ach = algo / sec / 1e9
            traffic, src = profile_figure(wname, 'hbm_bytes_per_read')
            busy, _ = profile_figure(wname, 'valu_busy_frac')
            out['roofline'] = {'bound': 'hbm', 'kernel': 'align (reverse-sweep launch + forward-sweep launch)',
                               'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                               # PMC figures cannot be measured inside this run: taken from a committed
                               # rocprofv3 summary of the same workload, scaled per read, and labelled
                               'traffic': traffic * n_reads if traffic else None, 'valu_busy': busy,
                               'from_profile': src,
                               'algorithmic_bytes_per_launch': algo, 'kernel_ms_per_launch': ms / launches,
                               'all_kernels_ms': per_kernel}
        elif launches and not is_api:
            # FP64 vector issue: lane-instructions executed (committed PMC count per read) / time, against
            # the measured issue rate of v_fma_f64 on all 1024 SIMDs
            sec = ms / 1000.0 / launches
            insts, src = profile_figure('cfg3_snps', 'valu_insts_per_read')
            traffic, _ = profile_figure('cfg3_snps', 'hbm_bytes_per_read')
            algo = dbatch.algorithmic_bytes_snp(stats['band_cells'])
            rl = {'bound': 'valu', 'kernel': kname, 'unit': 'Tlane-inst/s', 'peak': VALU_PEAK_TLANE,
                  'note': 'vector lane-instructions per second, in units of 1e12 (an FMA counts once, so these '
                          'are not flops); peak = 1 wave64 instruction / %.2f cycles (4 for FP64 and compare / select '
                          'forms, 2 for plain 32-bit integer ones, weighted by the hypothesis loop\'s mix) / SIMD at '
                          '2.4 GHz x 1024 SIMDs x 64 lanes' % VALU_CYCLES_PER_INST, 'from_profile': src,
                  'kernel_ms_per_launch': ms / launches, 'all_kernels_ms': per_kernel,
                  'traffic': traffic * n_reads if traffic else None, 'algorithmic_bytes_per_launch': algo,
                  'hbm_frac': algo / sec / 1e9 / HBM_PEAK_GBS}
            if insts:
                rl['achieved'] = insts * n_reads * 64 / sec / 1e12
                rl['frac'] = rl['achieved'] / VALU_PEAK_TLANE
                # `frac` is pipe utilisation: any wasted instruction raises it.  Work efficiency: executed
                # lane-instructions per row-cell update (4 R sweep rows + 3 R (2 k + 1) hypothesis rows,
                # each as wide as its band) against what the recurrences need (one table density
                # per two row-cells, a mixture + multiply-add on a wobble row, an emission product + two
                # multiply-adds on an emitting row, each value a (mantissa, exponent) pair: ~21)
                rows_per_base = 4 + 3 * (2 * model[0] + 1)
                per_cell = insts * 64.0 / (rows_per_base * stats['band_cells'] / n_reads)
                rl['lane_insts_per_row_cell_update'] = per_cell
                rl['floor_lane_insts_per_row_cell_update'] = 21.0
                rl['work_efficiency'] = rl['frac'] * 21.0 / per_cell
            else:
                rl['achieved'] = rl['frac'] = None
            out['roofline'] = rl
        elif is_api:
            out['kernels_ms_per_step'] = {k: v[0] / args.steps for k, v in timing.items() if v[1]}
        if wname == 'cfg2_align' and world == 1 and not args.no_e2e:
            # T_e2e: host arrays in -> host arrays out, every PCIe copy inside the timed region.
            # (a) a STREAM of batches through nvk_refine_alignment_submit / _wait, two in flight: the upload of
            #     batch k+1 and the download of batch k-1 run behind the kernels of batch k (csrc/pipeline.hip);
            # (b) ONE call of nvk_refine_alignment_batch: the batch in three growing chunks over three lanes —
            #     what a single call can hide is bounded by its first upload and a half-empty chip at its start.
            flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off,
                                             batch.context_before, batch.cb_off, batch.context_after,
                                             batch.ca_off, batch.anchors, batch.anc_off)
            ev_res = events.cpu().numpy()
            rs = dtw.RefineStream(km, bandwidth, mel, True)
            outs = [(np.zeros((dbatch.total_ref, 2), np.int32), np.zeros(n_reads, np.int32),
                     np.zeros(n_reads, np.int32)) for _ in range(2)]
            for i in range(6):   # warm-up: every lane once or twice (its staging, its 25 GB of spill), page-ins
                rs.wait(rs.submit(flat, out=outs[i % 2]))
            K = max(args.steps, 4)
            t1 = time.perf_counter()
            tickets = []
            for i in range(K):
                tickets.append(rs.submit(flat, out=outs[i % 2]))
                if i >= 1:
                    got = rs.wait(tickets[i - 1])
            got = rs.wait(tickets[-1])
            ts = time.perf_counter() - t1
            same = bool(np.array_equal(got[0], ev_res))
            algo = dbatch.algorithmic_bytes_align(stats['band_cells'])
            out['e2e'] = {'reads_per_s': n_reads * K / ts, 'ms_per_batch': 1000.0 * ts / K, 'batches': K,
                          'in_flight': 2, 'frac_of_value': (n_reads * K / ts) / out['value'],
                          'equals_resident_result': same,
                          'roofline': {'bound': 'hbm', 'achieved': algo / (ts / K) / 1e9, 'peak': HBM_PEAK_GBS,
                                       'unit': 'GB/s', 'frac': algo / (ts / K) / 1e9 / HBM_PEAK_GBS,
                                       'note': 'algorithmic bytes of the align kernel / whole-batch wall time, '
                                               'PCIe-inclusive'},
                          'includes': 'per batch: H2D of every input array (pageable host memory) + plan + align + '
                                      'D2H of events, status and tie flags; nvk_refine_alignment_submit/_wait'}
            dtw.refine_alignment_flat(flat, bandwidth, mel, km, True)
            t1 = time.perf_counter()
            ev1, _ = dtw.refine_alignment_flat(flat, bandwidth, mel, km, True)
            te = time.perf_counter() - t1
            out['e2e_single_call'] = {'reads_per_s': n_reads / te, 'ms': 1000.0 * te,
                                      'equals_resident_result': bool(np.array_equal(ev1, ev_res)),
                                      'includes': 'one nvk_refine_alignment_batch call: H2D + plan + align + D2H in '
                                                  'three overlapped chunks'}
        if not args.no_cpu_baseline and world == 1 and batch is not None:  # reported at N=1 only (contract)
            out['cpu_baseline'] = cpu_baseline(batch, model, bandwidth, mel, wname)
            out['gpu_over_cpu'] = out['value'] / world / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
