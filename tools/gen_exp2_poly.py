"""Generate the degree-11 polynomial for 2^f, f in [-0.5, 0.5], used by the sweep kernels
(nadavca_amd/csrc/xmath.h).  Chebyshev-node interpolation solved exactly in rational
arithmetic from 60-digit values, so the coefficients are correctly rounded doubles.
Prints C initialisers and the measured max relative error of the double Horner form."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math

getcontext().prec = 70
DEG = 11
LN2 = Decimal(2).ln()


def exp2(x):
    return (LN2 * x).exp()


def cos_dec(x):  # Taylor, x Decimal
    s, term, n = Decimal(0), Decimal(1), 0
    while abs(term) > Decimal(10) ** -65:
        s += term
        n += 2
        term = -term * x * x / ((n - 1) * n)
    return s


PI = Decimal('3.14159265358979323846264338327950288419716939937510582097494459230781640628620899')
nodes = [Decimal('0.5') * cos_dec(PI * (2 * j + 1) / (2 * (DEG + 1))) for j in range(DEG + 1)]
vals = [exp2(x) for x in nodes]
A = [[Fraction(x) ** k for k in range(DEG + 1)] + [Fraction(v)] for x, v in zip(nodes, vals)]
n = DEG + 1
for c in range(n):
    p = max(range(c, n), key=lambda r: abs(A[r][c]))
    A[c], A[p] = A[p], A[c]
    for r in range(n):
        if r != c:
            f = A[r][c] / A[c][c]
            A[r] = [a - f * b for a, b in zip(A[r], A[c])]
coef = [float(A[i][n] / A[i][i]) for i in range(n)]

worst = 0.0
for t in range(-4000, 4001):
    f = t / 8000.0
    p = coef[DEG]
    for k in range(DEG - 1, -1, -1):
        p = math.fma(p, f, coef[k]) if hasattr(math, 'fma') else p * f + coef[k]
    ref = exp2(Decimal(f))
    worst = max(worst, abs(float((Decimal(p) - ref) / ref)))
print('// max relative error of the double Horner evaluation on [-0.5,0.5]: %.3g' % worst)
print('static constexpr double EXP2_C[%d] = {' % n)
for c in coef:
    print('    %s,  // %r' % (c.hex(), c))
print('};')
