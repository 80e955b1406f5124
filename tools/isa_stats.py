"""Print per-kernel ISA statistics from the gfx950 assembly (dev tool)."""
import re, collections, subprocess, sys
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
s = open(path).read()
funcs = re.split(r'\n\s*\.globl\s+', s)
names = [f.split('\n', 1)[0].strip() for f in funcs[1:]]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
keys = ['global_load_dwordx4', 'global_store_dwordx4', 'global_load_dwordx2', 'global_load_dword', 'global_store_dword',
        'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_sqrt_f32',
        'v_div_scale_f32', 'v_div_fixup_f32', 'v_rcp_f32', 's_waitcnt', 'scratch_load_dword', 'scratch_store_dword']
for f, d in zip(funcs[1:], dem):
    if pat and not re.search(pat, d):
        continue
    g = lambda k: (re.search(r'; %s: (\d+)' % k, f) or [None, '?'])[1]
    hist = collections.Counter(re.sub(r'_e(32|64)$', '', m) for m in re.findall(r'^\s+([a-z_0-9]+)\s', f, re.M))
    print(d[:150])
    print('   vgpr', g('NumVgprs'), 'sgpr', g('TotalNumSgprs'), 'occ', g('Occupancy'), 'scratch', g('ScratchSize'),
          {k: hist[k] for k in keys if hist[k]}, 'total', sum(hist.values()))
