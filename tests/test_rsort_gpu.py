"""GPU: the library's own radix sort (csrc/rsort.hip, round 5 -- it replaced the hipCUB call of round 4) against torch's stable
sort: every length around the tile size, key widths that need one to four passes with both digit widths, duplicate-heavy keys
(stability), identity and given values, two sorts per launch, element count from device memory."""
import pytest
import torch

from nr3d_lib_amd import _hip as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TILE = 6144          # rsort::kTile


def _expect(keys, values, bits):
    k = keys.to(torch.int64) & 0xFFFFFFFF
    masked = k & ((1 << bits) - 1)
    order = torch.sort(masked, stable=True).indices
    v = values if values is not None else torch.arange(keys.numel(), device=keys.device, dtype=torch.int32)
    return keys[order], v[order]


def _rand_keys(n, bits, gen, dup=False):
    hi = 1 << min(bits, 31)
    k = torch.randint(0, 16 if dup else hi, (n,), generator=gen, dtype=torch.int64)
    if bits == 32:
        k = k | (torch.randint(0, 2, (n,), generator=gen, dtype=torch.int64) << 31)
    return (k & 0xFFFFFFFF).to(torch.int32) if bits < 32 else torch.where(k >= 1 << 31, k - (1 << 32), k).to(torch.int32)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, TILE - 1, TILE, TILE + 1, 3 * TILE + 17, 200_003, 1 << 20])
@pytest.mark.parametrize("bits", [1, 8, 9, 16, 17, 18, 19, 27, 32])
def test_sort_matches_stable_sort(n, bits):
    gen = torch.Generator().manual_seed(n * 37 + bits)
    keys = _rand_keys(n, bits, gen).to(DEV)
    ko, vo = H.sort_pairs_u32(keys, None, bits)
    ke, ve = _expect(keys, None, bits)
    assert torch.equal(ko, ke) and torch.equal(vo, ve)


@pytest.mark.parametrize("bits", [4, 17, 24])
def test_sort_is_stable_on_heavy_duplicates_and_carries_values(bits):
    gen = torch.Generator().manual_seed(bits)
    n = 5 * TILE + 123
    keys = _rand_keys(n, bits, gen, dup=True).to(DEV)
    vals = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), generator=gen, dtype=torch.int64).to(torch.int32).to(DEV)
    ko, vo = H.sort_pairs_u32(keys, vals, bits)
    ke, ve = _expect(keys, vals, bits)
    assert torch.equal(ko, ke) and torch.equal(vo, ve)
    # all keys equal: the identity
    ko, vo = H.sort_pairs_u32(torch.full((n,), 5, dtype=torch.int32, device=DEV), None, bits)
    assert torch.equal(vo, torch.arange(n, device=DEV, dtype=torch.int32))


def test_high_bits_beyond_the_sorted_range_are_carried_not_sorted():
    gen = torch.Generator().manual_seed(3)
    n = 50_000
    keys = _rand_keys(n, 31, gen).to(DEV)
    ko, vo = H.sort_pairs_u32(keys, None, 12)
    ke, ve = _expect(keys, None, 12)
    assert torch.equal(ko, ke) and torch.equal(vo, ve)


@pytest.mark.parametrize("n", [TILE + 5, 300_000])
def test_two_sorts_in_one_call(n):
    gen = torch.Generator().manual_seed(n)
    k0, k1 = _rand_keys(n, 17, gen).to(DEV), _rand_keys(n, 15, gen).to(DEV)
    v1 = torch.randperm(n, generator=gen).to(torch.int32).to(DEV)
    (o0, o1), (p0, p1) = H.sort_pairs_u32([k0, k1], [None, v1], 17)
    e0, q0 = _expect(k0, None, 17)
    e1, q1 = _expect(k1, v1, 17)
    assert torch.equal(o0, e0) and torch.equal(p0, q0) and torch.equal(o1, e1) and torch.equal(p1, q1)


@pytest.mark.parametrize("n_live", [0, 1, 777, TILE, 2 * TILE + 1, 100_000])
def test_count_from_device_memory(n_live):
    gen = torch.Generator().manual_seed(n_live + 1)
    n = 100_000
    keys = _rand_keys(n, 18, gen).to(DEV)
    vals = torch.randperm(n, generator=gen).to(torch.int32).to(DEV)
    nd = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    ko, vo = H.sort_pairs_u32(keys, vals, 18, n_dev=nd)
    ke, ve = _expect(keys[:n_live], vals[:n_live], 18)
    assert torch.equal(ko[:n_live], ke) and torch.equal(vo[:n_live], ve)


def test_inputs_are_left_untouched():
    gen = torch.Generator().manual_seed(9)
    keys = _rand_keys(40_000, 27, gen).to(DEV)
    vals = torch.randperm(40_000, generator=gen).to(torch.int32).to(DEV)
    k0, v0 = keys.clone(), vals.clone()
    H.sort_pairs_u32(keys, vals, 27)
    assert torch.equal(keys, k0) and torch.equal(vals, v0)
