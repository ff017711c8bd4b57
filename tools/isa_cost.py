#!/usr/bin/env python3
"""Static issue-cost model of a gfx950 kernel's ISA (hipcc -S output), per basic block.

Cycle classes per wave64 VALU instruction, from profiles/r01_microbench_valu_rates.txt (MI355X, SIMD-32):
  full (2 cycles):   v_mul/add/sub_f32, v_fma/fmac/fmaak/fmamk on distinct VGPR sources and no SGPR source (measured 2.3-2.4 at 8 waves per SIMD, round 6),
                     v_and/or/xor_b32, v_add/sub_u32, v_mov_b32 — inline constants and literals are free
  half (4 cycles):   everything else (v_cvt, v_min/max, v_cmp, v_cndmask, shifts, v_bfi, v_mad_*, SDWA forms, an fma that reads one VGPR twice,
                     v_pk_* (two elements), any VALU with an SGPR source operand)
  trans (8 cycles):  v_rcp/rsq/sqrt/exp/log/sin/cos
usage: isa_cost.py file.s [kernel-substring]      prints per-block instruction and cycle counts in program order
"""
import re, sys, collections

FULL = {'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_add_u32', 'v_sub_u32',
        'v_subrev_u32', 'v_mov_b32', 'v_fma_f32', 'v_fmac_f32', 'v_fmaak_f32', 'v_fmamk_f32', 'v_mac_f32', 'v_mad_f32', 'v_not_b32'}
TRANS = ('v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_exp_', 'v_log_', 'v_sin_', 'v_cos_')


def classify(op, args):
    """-> (kind, cycles) for one instruction"""
    if op.startswith('v_'):
        base = op
        for suf in ('_e32', '_e64', '_sdwa', '_dpp'):
            if base.endswith(suf):
                base = base[:-len(suf)]
        if base.startswith(TRANS):
            return 'valu', 8
        if op.endswith('_sdwa') or op.endswith('_dpp') or base.startswith('v_pk_'):
            return 'valu', 4
        if base in ('v_readlane_b32', 'v_readfirstlane_b32', 'v_writelane_b32'):
            return 'valu', 4
        if base in FULL:
            srcs = args[1:]
            if base in ('v_fmac_f32', 'v_mac_f32'):
                srcs = args[1:] + [args[0]]
            nv = len([a for a in srcs if re.match(r'^[-|]*v\d+|^[-|]*v\[', a.strip('|-'))])
            vregs = set(a.strip('|-') for a in srcs if re.match(r'^v(\d+|\[)', a.strip('|-')))
            has_s = any(re.match(r'^(s\d+|s\[|vcc|exec|ttmp|m0)', a.strip('|-')) for a in srcs)
            if has_s:
                return 'valu', 4
            # (round 6, profiles/r06_microbench_cycles.txt: a v_fma_f32 on three DISTINCT VGPRs issues at the full rate, 2.37 cycles at 8 waves; one that names the
            #  same VGPR twice does not, 3.74 — a register-bank effect, not the three-operand encoding)
            if nv > len(vregs):
                return 'valu', 4
            return 'valu', 2
        return 'valu', 4
    if op.startswith('s_'):
        if op.startswith(('s_load', 's_buffer_load', 's_memtime', 's_memrealtime')):
            return 'smem', 0
        if op.startswith('s_waitcnt'):
            return 'wait', 0
        return 'salu', 0
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem', 0
    if op.startswith('ds_'):
        return 'lds', 0
    return 'other', 0


def parse(path, want=None):
    kernels = collections.OrderedDict()
    cur = None
    block = None
    for line in open(path):
        s = line.split(';')[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', s)
        if m:
            name = m.group(1)
            if not name.startswith('.L'):
                cur = kernels.setdefault(name, collections.OrderedDict())
                block = cur.setdefault('entry', [])
            elif cur is not None:
                block = cur.setdefault(name, [])
            continue
        if cur is None or s.lstrip().startswith('.'):
            continue
        parts = s.strip().split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
        block.append((op, args, s.strip()))
    return kernels


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    verbose = len(sys.argv) > 3
    for kname, blocks in parse(path).items():
        if want and want not in kname:
            continue
        if sum(len(b) for b in blocks.values()) < 20:
            continue
        print(f'== {kname[:110]}')
        tot = collections.Counter()
        for bname, ins in blocks.items():
            c = collections.Counter()
            cyc = 0
            for op, args, _ in ins:
                kind, cy = classify(op, args)
                c[kind] += 1
                cyc += cy
            tot.update(c)
            tot['cycles'] += cyc
            term = ins[-1][2] if ins else ''
            br = [i[2] for i in ins if i[0].startswith(('s_cbranch', 's_branch'))]
            print(f'  {bname:12s} n={len(ins):4d} valu={c["valu"]:4d} cyc={cyc:5d} salu={c["salu"]:3d} smem={c["smem"]:2d} vmem={c["vmem"]:2d} lds={c["lds"]:2d} wait={c["wait"]:2d}  {" | ".join(br)[:70]}')
            if verbose:
                for op, args, s in ins:
                    print(f'        {classify(op, args)[1]}  {s}')
        print('  total', dict(tot))


if __name__ == '__main__':
    main()
