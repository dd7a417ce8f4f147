"""Nucleotide tables used across the package.  Same four names as the reference module
(/root/reference/nadavca/alphabet.py) so that code written against it keeps working:
``alphabet`` (index -> base), ``inv_alphabet`` (base -> index), ``complement`` (base -> base) and
``numerical_complement`` (index -> index of the complementary base, i.e. 3 - index)."""

_ORDER = 'ACGT'
_PAIRS = ('AT', 'CG')

alphabet = list(_ORDER)
inv_alphabet = dict(zip(_ORDER, range(len(_ORDER))))
complement = {}
for _x, _y in _PAIRS:
    complement[_x], complement[_y] = _y, _x
numerical_complement = {inv_alphabet[b]: inv_alphabet[complement[b]] for b in _ORDER}

assert all(numerical_complement[i] == len(_ORDER) - 1 - i for i in range(len(_ORDER)))
