"""CPU tests of the fused GEMM's host-side planning (pure arithmetic behind hqq_hip_gemm_plan / *_workspace_bytes / the routing hint):
every plan covers K exactly once per tile, workspace sizes match the plan, hybrids keep whole rounds whole."""
import ctypes
import itertools

import pytest


@pytest.fixture(scope="module")
def L():
    from hqq_amd import _C
    return _C.lib()


SHAPES = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672),
          (256, 128), (72, 1280), (5120, 2048)]
MS = [1, 17, 64, 65, 100, 128, 129, 256, 300, 512, 640, 641, 768, 1024, 1025, 1536, 2048, 3000, 4096, 8192, 65536]


@pytest.mark.parametrize("nbits,dtype", [(4, 1), (8, 1), (2, 1), (4, 2)])
def test_plans_are_consistent(L, nbits, dtype):
    out = (ctypes.c_int * 8)()
    per = 8 // nbits
    for (N, K), M in itertools.product(SHAPES, MS):
        if N % per or (N // per) % 4:
            continue
        for opts in (0, 64, 128, 192, 256, 2 << 24, 192 | (3 << 24)):
            rc = L.hqq_hip_gemm_plan(nbits, M, N, K, 64, dtype, opts, out)
            assert rc == 0, (nbits, M, N, K, opts)
            nw, bm, n_tiles, m_tiles, ks, kps, full, wgs = list(out)
            nk = K // 64
            assert nw in (4, 8) and bm in (128, 256) and not (bm == 256 and (nw != 8 or nbits == 2))
            assert n_tiles * 16 * nw >= N // per > (n_tiles - 1) * 16 * nw and m_tiles * bm >= M > (m_tiles - 1) * bm
            assert ks >= 1 and kps % 2 == 0 and ks * kps >= nk > (ks - 1) * kps          # the splits cover K once, none is empty
            tiles = n_tiles * m_tiles
            assert 0 <= full < tiles and wgs == (tiles if ks == 1 else full + (tiles - full) * ks)
            if full:
                assert full % 256 == 0 and ks > 1 and (tiles - full) * ks <= 256 and not (opts & 256)
            ws = L.hqq_hip_gemm_workspace_bytes(nbits, M, N, K, 64, dtype, opts)
            want = 0 if ks == 1 else 256 * 1024 + ks * (tiles - full) * bm * 16 * nw * per * 4
            assert ws == want, (nbits, M, N, K, opts, list(out), ws, want)
            if (opts >> 24):
                assert ks <= (opts >> 24)
            # forced tile shapes are honoured
            if opts & 192 == 64: assert (nw, bm) == (4, 128)
            if opts & 192 == 128: assert (nw, bm) == ((8, 128) if nbits != 2 else (4, 128))   # (2-bit layers always take the 4-wave tile: csrc/gemm_pipe.hip GD_2BIT_ONE_WAVE_PER_SIMD)
            if opts & 192 == 192 and nbits != 2: assert (nw, bm) == (8, 256)


def test_routing_hint_and_unsupported_shapes(L):
    out = (ctypes.c_int * 8)()
    assert L.hqq_hip_gemm_plan(4, 128, 4096, 4096, 128, 1, 0, out) == -4        # group_size 128: the output-tile kernels serve it
    assert L.hqq_hip_gemm_plan(3, 128, 4096, 4096, 64, 1, 0, out) == -4
    assert L.hqq_hip_gemm_plan(4, 128, 4096, 4096 + 64, 64, 1, 0, out) == -4    # K % 128 != 0
    assert L.hqq_hip_gemm_plan(4, 128, 4096, 4096, 64, 1, 32, out) == -4        # HQQ_OPT_GEMM_CLASSIC
    assert L.hqq_hip_gemm_plan(4, 128, 4096, 4096, 64, 0, 0, out) == -4         # fp32 compute dtype
    for M in (1, 16, 17, 64, 65, 640):
        assert L.hqq_hip_forward_prefers_fused(4, M, 4096, 4096, 64, 1) == 1
    assert L.hqq_hip_forward_prefers_fused(4, 32, 4096, 384, 64, 1) == 1        # 17..64 rows outside the skinny kernel: the pipelined GEMM
    assert L.hqq_hip_forward_prefers_fused(4, 32, 4096, 4096 + 64, 64, 1) == 0   # ... which needs K % 128 == 0
    assert L.hqq_hip_forward_prefers_fused(4, 2048, 4096, 4096, 64, 1) == 1 and L.hqq_hip_forward_prefers_fused(4, 3072, 4096, 4096, 64, 1) == 0
