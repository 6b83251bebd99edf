#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3p; mkdir -p $OUT; cd $ROOT
FL="0 67108864 134217728 201326592 268435456 335544320"
for f in $FL; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family fd_dw" $OUT/bf16_$f.txt | head -5; done
for k in "fd_dwconv_train<" "fd_dw_bwd"; do
echo "== $k: flags $FL"
paste <(grep -E "$k" $OUT/bf16_0.txt | grep -v family | awk '{print $2, $3}') $(for f in 67108864 134217728 201326592 268435456 335544320; do echo "<(grep -E \"$k\" $OUT/bf16_$f.txt | grep -v family | awk '{print \$3}')"; done | sed 's/^/ /' | tr -d '\n' | xargs -0 -I{} echo {} > /dev/null; echo) 2>/dev/null
done
python - <<'P'
import re,sys,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3p'
fl="0 67108864 134217728 201326592 268435456 335544320".split()
rows={}
for f in fl:
    for ln in open('%s/bf16_%s.txt'%(out,f)):
        m=re.match(r'\s*(\d+)\s+(\S+)\s+([\d.]+)\s+(fd_dw\S+)',ln)
        if m: rows.setdefault((int(m.group(1)),m.group(2),m.group(4).split('<')[0]),{})[f]=float(m.group(3))
for k in sorted(rows): print('%-16s %-16s'%(k[1],k[2]),' '.join('%7.1f'%rows[k].get(f,-1) for f in fl))
P
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_batch" 2>&1 | tail -n 3
