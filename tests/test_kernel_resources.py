"""Compile-time properties of the hand-scheduled kernels that their performance rests on (no GPU needed: hipcc
cross-compiles gfx950): the persistent layer kernel and its tail / prologue modes must not touch scratch - a scratch
reload is vector memory, completes in order behind the weight-stream DMA and drains the look-ahead - and
must fit one wave per SIMD; the tile GEMMs must fit two blocks of 512 threads per CU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_layer_kernels_have_no_scratch(tmp_path):
    out = tmp_path / 'gemm_bf16.s'
    src = os.path.join(ROOT, 'ddp_amd', 'csrc', 'ddp_gemm_bf16.hip')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-x', 'hip', src,
                    '--cuda-device-only', '-S', '-o', str(out)], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
        body = m.group(2)
        kernels[m.group(1)] = (int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)),
                               int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)))
    layer = {k: v for k, v in kernels.items() if 'k_layer' in k}
    gemm = {k: v for k, v in kernels.items() if 'k_gemm' in k}
    assert len(layer) >= 6 and len(gemm) >= 8
    for name, (scratch, vgpr) in layer.items():
        assert scratch == 0, f'{name}: {scratch} B of scratch'
        assert vgpr <= 512
    for name, (scratch, vgpr) in gemm.items():
        assert vgpr <= 256, f'{name}: {vgpr} registers (two 512-thread blocks per CU need <= 256)'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_layer_tail_kernels_have_no_scratch(tmp_path):
    """k_layer MODE 6 (last layer of a step + seg tail, its own translation unit): the whole layer kernel plus a tail in one
    function body - the allocator must still keep every phase in the 512 registers (two spill sources were removed by hand:
    the next q's store addresses derived at the top of the tile, and a second exit condition that kept the old fragments alive)"""
    out = tmp_path / 'layer_tail.s'
    src = os.path.join(ROOT, 'ddp_amd', 'csrc', 'ddp_layer_tail.hip')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-x', 'hip', src,
                    '--cuda-device-only', '-S', '-o', str(out)], check=True, capture_output=True, timeout=600)
    found = 0
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', out.read_text(), re.S):
        if 'k_layer' not in m.group(1):
            continue
        found += 1
        body = m.group(2)
        assert int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)) == 0, m.group(1)
        assert int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)) <= 512
    # MODE 6: {1, 2, 3, 4 chunks of 64 classes} x {plain, non-temporal, teacher-forced}; MODE 7, 8 (bev tail), 9 (depth tail): {plain, non-temporal}
    assert found == 18


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_gather_fits_four_blocks_per_cu(tmp_path):
    """the LDS-staged gather overlaps fill and taps ACROSS blocks (round 6: FOUR 512-thread blocks per CU = 8 waves per SIMD): no
    scratch, <= 64 registers, and a window + bookkeeping of at most 160 KiB / 4 of LDS"""
    out = tmp_path / 'kernels.s'
    src = os.path.join(ROOT, 'ddp_amd', 'csrc', 'ddp_kernels.hip')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-x', 'hip', src,
                    '--cuda-device-only', '-S', '-o', str(out)], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    found = 0
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
        if 'k_msda_gather_lds' not in m.group(1):
            continue
        found += 1
        body = m.group(2)
        assert int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)) == 0
        assert int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)) <= 64
        assert int(re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', body).group(1)) <= 160 * 1024 // 4
    assert found == 2          # the SB-output and the fp32-fragment-output instantiation
