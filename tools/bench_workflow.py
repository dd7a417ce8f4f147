"""End-to-end timing of the host workflow around the kernels: get_refined_alignments,
estimate_probabilities (consensus, independent=False) and estimate_probabilities_independent on
synthetic reads sampled from a genome (the synthetic aligner returns the true base mapping, standing
in for BWA).  `python tools/bench_workflow.py N [profile]`."""
import sys, time, cProfile, pstats, io
import numpy as np
import yaml
sys.path.insert(0, '.')
from nadavca_amd import synthetic, defaults, kmer_model as km, _lib
from nadavca_amd.estimator import ProbabilityEstimator
from nadavca_amd.alignment import ApproximateAligner
from nadavca_amd.read import Read

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
prof = len(sys.argv) > 2 and sys.argv[2] == 'profile'
config = yaml.safe_load(open(defaults.CONFIG_FILE))
model = km.load_kmer_model(defaults.KMER_MODEL_FILE)
arrays = synthetic.load_model_arrays()
genome = synthetic.make_genome(10000, 3)
t0 = time.perf_counter()
specs = synthetic.make_read_specs(n_reads, genome, arrays, seed=17)
reads = synthetic.reads_from_specs(specs)
Read.normalize_reads(reads)
print('built %d reads in %.1f s' % (n_reads, time.perf_counter() - t0))
aligner = synthetic.make_synthetic_aligner(ApproximateAligner, genome)
est = ProbabilityEstimator(model, aligner, config)
ctx = _lib.default_context()

def timed(name, fn):
    fn()  # warm-up (workspaces, first-touch)
    ctx.timing_reset(); ctx.timing_enable(True)
    pr = cProfile.Profile() if prof else None
    t = time.perf_counter()
    if pr: pr.enable()
    out = fn()
    if pr: pr.disable()
    dt = time.perf_counter() - t
    ctx.timing_enable(False)
    kern = sum(ms for ms, n in ctx.timing_read().values())
    print('%-38s %8.1f ms wall, %7.1f ms in kernels, %8.0f reads/s' % (name, dt * 1e3, kern, n_reads / dt))
    if pr:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(s.getvalue()[:3500])
    return out

timed('get_refined_alignments', lambda: est.get_refined_alignments(reads))
timed('estimate_probabilities (consensus)', lambda: est.estimate_probabilities(genome, reads))
timed('estimate_probabilities_independent', lambda: est.estimate_probabilities_independent(genome, reads))


# ---- align_signal's loop: device-resident (the product) vs the same rounds through the host ----------
def host_loop():
    from scipy.stats import linregress
    res = est.get_refined_alignments(reads)
    for r in range(3):
        if r % 2 == 0:
            for read, (apx, al) in zip(reads, res):
                bases = [{'A': 0, 'C': 1, 'G': 2, 'T': 3}[x] for x in apx.reference_part]
                expected = np.array(model.get_expected_signal(bases, [], []))
                means = [np.mean(read.normalized_signal[s:e]) for _, s, e in al]
                slope, intercept = linregress(expected, means)[:2]
                read.normalized_signal = (read.normalized_signal - intercept) / slope
        else:
            res = est.get_refined_alignments(reads)
    return res


def fresh():
    Read.normalize_reads(reads)


fresh()
timed('align+renorm loop, rounds on the host', host_loop)
fresh()
timed('align+renorm loop, device-resident', lambda: est.refine_and_renormalize(reads, 3))
t = time.perf_counter(); Read.normalize_reads(reads); th = time.perf_counter() - t
Read.normalize_reads_device(reads)
t = time.perf_counter(); Read.normalize_reads_device(reads); td = time.perf_counter() - t
print('normalize_reads (one group, %d samples): host %.1f ms, device %.1f ms (with H2D/D2H)' % (
    sum(len(r.raw_signal) for r in reads), th * 1e3, td * 1e3))
t = time.perf_counter()
for r in reads:
    Read.normalize_reads([r])
th = time.perf_counter() - t
t = time.perf_counter(); Read.normalize_reads_device(reads, per_read=True); td = time.perf_counter() - t
print('normalize_reads per read: host %.1f ms, device %.1f ms (with H2D/D2H)' % (th * 1e3, td * 1e3))
