#!/bin/bash
# A/B of train plan flags (and library variants: <flags>@<path of a tools/build_variant.py build>), per-layer lines side by side:
# gpurun --timeout 600 -- bash tools/gpu_ab.sh <dtype> <flags[@lib]...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ab; mkdir -p $OUT; cd $ROOT
DT=$1; shift; FL="$@"
for f in $FL; do fl=${f%%@*}; lib=; [ "$f" != "$fl" ] && lib="--lib ${f#*@}"; tag=$(echo $f | tr "/@" "__"); timeout 200 python tools/train_layer_times.py --dtype $DT --plan-flags $fl $lib > $OUT/${DT}_$tag.txt 2>&1; grep -E "plan flags" $OUT/${DT}_$tag.txt; done
FL="$FL" DT=$DT python - <<'P'
import re,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/ab'
fl=os.environ['FL'].split(); dt=os.environ['DT']
rows={}; fam={}
for f in fl:
    for ln in open('%s/%s_%s.txt'%(out,dt,f.replace('/','_').replace('@','_'))):
        m=re.match(r'\s*(-?\d+)\s+(\S+)\s+([\d.]+)\s+(fd_\w+)',ln)
        if m and int(m.group(1))>=0:
            k=(int(m.group(1)),m.group(2)); rows.setdefault(k,{}).setdefault(f,[]).append((m.group(4),float(m.group(3))))
        m=re.match(r'\s*family (\S+)\s+(\d+) launches\s+([\d.]+)',ln)
        if m: fam.setdefault(m.group(1),{})[f]=float(m.group(3))
print('flags:',' '.join(fl))
for k in sorted(rows):
    print('%-16s'%k[1],' | '.join(' '.join('%s=%.1f'%(n.replace('fd_','')[:14],t) for n,t in rows[k].get(f,[]) if 'finalize' not in n) for f in fl))
for k in sorted(fam): print('family %-30s'%k,' '.join('%8.1f'%fam[k].get(f,0) for f in fl))
P
