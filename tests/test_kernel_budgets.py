"""Register budgets of the hot kernels (no GPU: the compiler's metadata for gfx950, tools/kernel_resources.py).

Occupancy on CDNA4 moves in steps — <= 128 VGPRs four waves per SIMD, <= 168 three, above two — and nothing fails when
a kernel drifts across one: it just gets slower.  Round 5 found two such drifts after the fact (the SH forward at 173
registers instead of 158: +20 us per frame; the fixed-point per-gaussian backward at 172 instead of 160: +190 us), so the
budgets the kernels were tuned at are asserted here, together with "no scratch" for every kernel of the frame's hot
path (a spill reload waits on vmcnt(0), i.e. on every gather and atomic in flight)."""
import shutil
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'tools'))

BUDGETS = {
  # kernel (demangled prefix) : (file, max VGPRs)
  'ms::raster_bwd_scan_kernel<16, false, 1, false, false>': ('raster_bwd_scan.hip', 128),
  'ms::raster_bwd_scan_kernel<16, true, 1, false, false>': ('raster_bwd_scan.hip', 128),
  'ms::raster_bwd_scan_kernel<32, false, 1, false, false>': ('raster_bwd_scan.hip', 128),
  'ms::raster_bwd_scan_kernel<16, false, 1, false, true>': ('raster_bwd_scan.hip', 128),
  'ms::raster_fwd_f32x3_kernel<16, false, false, false>': ('raster_fast.hip', 64),      # + 20 480 bytes of LDS: eight workgroups per CU
  'ms::raster_fwd_f32x3_kernel<16, true, false, false>': ('raster_fast.hip', 72),
  'ms::sh_fwd_rows_deg3_kernel<false>': ('sh.hip', 160),
  'ms::gaussian_bwd_kernel<float, 3, true, false>': ('gaussian_bwd.hip', 168),
  'ms::gaussian_bwd_kernel<float, 3, true, true>': ('gaussian_bwd.hip', 168),
  'ms::project_fwd_kernel<float>': ('projection.hip', 64),
  'ms::tile_count_direct_kernel<float>': ('mapper.hip', 64),
  'ms::tile_emit_direct_kernel<float>': ('mapper.hip', 64),
  'ms::tile_depth_sort_kernel<4>': ('tile_sort.hip', 64),
  'ms::radix_downsweep_kernel<unsigned long, ms::PlainPairs<unsigned long> >': ('scan_sort.hip', 168),
}


LDS_BUDGETS = {
  'ms::raster_fwd_f32x3_kernel<16, false, false, false>': 20480,     # 8 x 20 480 = the CU's 160 KB
  'ms::raster_bwd_scan_kernel<16, false, 1, false, false>': 40960,   # 4 workgroups per CU
  'ms::radix_downsweep_kernel<unsigned long, ms::PlainPairs<unsigned long> >': 54608,   # 3 workgroups per CU
}


@pytest.mark.skipif(shutil.which('hipcc') is None and not Path('/opt/rocm/bin/hipcc').exists(), reason="no hipcc")
def test_hot_kernels_stay_inside_their_register_budgets():
  import kernel_resources as kr
  from concurrent.futures import ThreadPoolExecutor
  files = sorted({f for f, _ in BUDGETS.values()})
  with ThreadPoolExecutor(len(files)) as pool:
    tables = dict(zip(files, pool.map(lambda f: kr.resources(kr.SRC / f), files)))
  problems = []
  for kernel, (f, max_vgpr) in BUDGETS.items():
    match = [(name, r) for name, r in tables[f].items() if name.replace('void ', '').startswith(kernel + '(')]
    assert len(match) == 1, (kernel, [n for n in tables[f] if kernel.split('<')[0] in n][:6])
    r = match[0][1]
    if r['vgpr'] > max_vgpr:
      problems.append(f"{kernel}: {r['vgpr']} VGPRs > {max_vgpr}")
    if r.get('scratch', 0) != 0:
      problems.append(f"{kernel}: {r['scratch']} bytes of scratch")
    if kernel in LDS_BUDGETS and r.get('lds', 0) > LDS_BUDGETS[kernel]:
      problems.append(f"{kernel}: {r['lds']} bytes of LDS > {LDS_BUDGETS[kernel]}")
  assert not problems, problems
