"""The inverse of Thomas Wang's 64-bit hash and the reverse complement of a 2-bit k-mer: helpers of the tests that MAKE k-mers
with a wanted hash (tests/test_gpu_sketch.py::test_kmers_whose_hash_has_32_zero_bits_behind_the_index); checked against the
Python oracle on the CPU by tests/test_oracle.py::test_wang_hash_inverse."""
_M64 = (1 << 64) - 1


def unwang(h):
    """the inverse of Thomas Wang's 64-bit hash (every step is a bijection of 64-bit words)"""
    h = (h * pow((1 << 31) + 1, -1, 1 << 64)) & _M64          # key += key << 31
    h ^= h >> 28
    h ^= h >> 56                                              # key ^= key >> 28
    h = (h * pow(21, -1, 1 << 64)) & _M64                     # key *= 21
    h ^= (h >> 14) ^ (h >> 28) ^ (h >> 42) ^ (h >> 56)        # key ^= key >> 14
    h = (h * pow(265, -1, 1 << 64)) & _M64                    # key *= 265
    h ^= (h >> 24) ^ (h >> 48)                                # key ^= key >> 24
    return ((h + 1) * pow((1 << 21) - 1, -1, 1 << 64)) & _M64  # key = ~key + (key << 21) = key * (2^21 - 1) - 1


def revcomp(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (x & 3))
        x >>= 2
    return r
