// Host build of nadavca_amd/csrc/splfit.h for tests/test_splfit_cpu.py (g++ -O2 -ffp-contract=off -shared):
// the same restatement the device kernel runs, compared there with scipy.interpolate.splrep.
#include "../../nadavca_amd/csrc/splfit.h"

extern "C" int splfit_host_cubic_first_pass(const double *x, const double *y, long long m, double *t8, double *c8) {
  return splfit::cubic_first_pass(x, y, (int64_t)m, t8, c8);
}
