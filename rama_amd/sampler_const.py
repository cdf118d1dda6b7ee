"""The constant draw of the reference's sampler, derived in-tree from the published algorithms.

`Device::sample` re-seeds its generator on EVERY call -- `ChaCha20Rng::seed_from_u64(100)` on the CPU
backend (engine/src/device/cpu.rs:161-162), `seed_from_u64(10)` on the CUDA backend (gpu.rs:151-152) --
and `sample_top_q` draws once (`rng.gen::<f32>()`, engine/src/transformer/infer.rs:75), so the uniform
variate is the same number for every token:

  * rand_core 0.6 `SeedableRng::seed_from_u64`: a PCG32 stream (multiplier 6364136223846793005, increment
    11634580027462260723; state advanced first, output = xorshift-high rotated right) fills the 32-byte
    seed four little-endian bytes at a time;
  * rand_chacha `ChaCha20Rng::from_seed`: that seed is the ChaCha key, block counter 0, stream 0; the first
    `next_u32()` is word 0 of keystream block 0 (20 rounds, RFC 7539's quarter round and constants);
  * rand 0.8 `Standard` for f32: `(next_u32() >> 8) as f32 * 2^-24`.

The crates are `*`-versioned in the reference (engine/Cargo.toml:15,17) and no Rust toolchain exists here, so
the value is derived, not observed; tests/test_sampler_const.py checks the ChaCha core against RFC 7539's
zero-key block and pins the two constants every host mirror uses.
"""
from __future__ import annotations

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def pcg32_seed_bytes(state: int, n_bytes: int = 32) -> bytes:
    """rand_core::SeedableRng::seed_from_u64's expansion of a u64 into a seed"""
    out = bytearray()
    for _ in range(n_bytes // 4):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        out += x.to_bytes(4, "little")
    return bytes(out)


def _rotl(v: int, c: int) -> int:
    return ((v << c) & M32) | (v >> (32 - c))


def chacha20_block(key: bytes, counter: int = 0, stream: int = 0) -> list[int]:
    """the 16 output words of one ChaCha20 block: 64-bit block counter in words 12-13, 64-bit stream id in 14-15 (rand_chacha's layout;
    with both zero it is RFC 7539's block 0 under a zero nonce)"""
    assert len(key) == 32
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    st += [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
    st += [counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + st[i]) & M32 for i in range(16)]


def first_word(seed: int) -> int:
    return chacha20_block(pcg32_seed_bytes(seed))[0]


def first_f32(seed: int) -> float:
    """ChaCha20Rng::seed_from_u64(seed).gen::<f32>() -- exactly representable in fp32 (24 significant bits)"""
    return (first_word(seed) >> 8) * 2.0 ** -24


# the two draws the reference can make (literals kept beside their derivation; tests/test_sampler_const.py recomputes them)
TOPP_U_CPU = 0.2721174359321594      # seed 100, cpu.rs:161-162 -- the parity target of this backend
TOPP_U_CUDA = 0.03743588924407959    # seed 10, gpu.rs:151-152
