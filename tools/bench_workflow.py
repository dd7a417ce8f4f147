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
