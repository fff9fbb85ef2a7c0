"""Instruction mix of the render kernels from the compiler's ISA listing (verdict r1 item 4).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -save-temps -c gendr_amd/csrc/gendr_capi.hip
    python tools/isa_mix.py gendr_capi-hip-amdgcn-amd-amdhsa-gfx950.s [substring of the kernel name ...]

Static counts per kernel: f32 VALU, f64 VALU (incl. conversions to / from f64), transcendental (v_exp/log/rcp/rsq/sqrt/sin/cos),
integer / move / compare VALU, cross-lane (readlane, dpp, permute), SALU, SMEM, VMEM loads / stores / atomics, LDS, waits,
scratch (spill) accesses -- and the registers / scratch the kernel descriptor asks for."""
import collections
import re
import sys


def classify(op):
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')):
        return 'wait/nop'
    if op.startswith(('s_load', 's_buffer_load', 's_store', 's_dcache', 's_memtime')):
        return 'smem'
    if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc', 's_getpc', 's_call')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('scratch_'):
        return 'scratch'
    if op.startswith(('global_atomic', 'flat_atomic', 'buffer_atomic')):
        return 'vmem_atomic'
    if op.startswith(('global_load', 'flat_load', 'buffer_load')):
        return 'vmem_load'
    if op.startswith(('global_store', 'flat_store', 'buffer_store')):
        return 'vmem_store'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane', 'v_mov_b32_dpp')) or '_dpp' in op:
        return 'crosslane'
    if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op):
        return 'trans_f64' if op.endswith('f64') else 'trans_f32'
    if 'f64' in op:
        return 'valu_f64'
    if re.match(r'v_(pk_)?(add|sub|mul|fma|mac|mad|max|min|fmac|fmaak|fmamk|ldexp|frexp|div_scale|div_fmas|div_fixup|trunc|floor|ceil|rndne|fract|cvt|med3|max3|min3|cmp\w*|cmpx\w*)_.*f32', op) or op.endswith('_f32') or '_f32_' in op:
        return 'valu_f32'
    if op.startswith('v_'):
        return 'valu_int/mov/cmp'
    return 'other'


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    kernels = collections.OrderedDict()
    meta = {}
    cur = None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m and 'gendr' in line:
            cur = m.group(1)
            kernels[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            cur = None
            continue
        m = re.match(r'^\t([a-z_0-9]+)', line)
        if m and not line.startswith('\t.'):
            kernels[cur][classify(m.group(1))] += 1
            kernels[cur]['op:' + m.group(1)] += 1
    # register / scratch metadata
    for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)',
                         open(path).read(), re.S):
        meta[m.group(1)] = dict(scratch=int(m.group(2)), sgpr=int(m.group(3)), vgpr=int(m.group(4)))
    cats = ['valu_f32', 'valu_f64', 'trans_f32', 'trans_f64', 'valu_int/mov/cmp', 'crosslane', 'salu', 'branch', 'smem',
            'vmem_load', 'vmem_store', 'vmem_atomic', 'lds', 'scratch', 'wait/nop', 'other']
    print('%-110s %s  vgpr sgpr scratch' % ('kernel', ' '.join('%9s' % c[:9] for c in cats)))
    for k, c in kernels.items():
        if want and not any(w in k for w in want):
            continue
        md = meta.get(k, {})
        print('%-110s %s  %4s %4s %4s' % (k[:110], ' '.join('%9d' % c[x] for x in cats), md.get('vgpr', '?'), md.get('sgpr', '?'), md.get('scratch', '?')))
        if '--ops' in sys.argv:
            for op, n in sorted(((o[3:], n) for o, n in c.items() if o.startswith('op:')), key=lambda t: -t[1])[:40]:
                print('      %-28s %d' % (op, n))


if __name__ == '__main__':
    main()
