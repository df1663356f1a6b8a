# usage (anywhere hipcc is, no GPU needed): bash tools/kernel_resources.sh > profiles/r04_kernel_resources.md
# hipcc -Rpass-analysis=kernel-resource-usage over every source of libltrx: VGPRs / AGPRs / scratch / spills / LDS per kernel
R=$(cd $(dirname $0)/.. && pwd)
echo "# kernel resources of libltrx.so (gfx950), hipcc -O3 -Rpass-analysis=kernel-resource-usage"
echo
echo "| source | kernel | VGPRs | AGPRs | scratch B/lane | VGPR spills | SGPR spills | occupancy waves/SIMD | static LDS B |"
echo "|---|---|---|---|---|---|---|---|---|"
for f in $R/allrank_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | \
  python3 -c "
import re,sys,subprocess
src='$(basename $f)'
cur=None
rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur={'name':m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key,pat in (('v',r' VGPRs: (\d+)'),('a',r'AGPRs: (\d+)'),('s',r'ScratchSize \[bytes/lane\]: (\d+)'),('vs',r'VGPRs Spill: (\d+)'),('ss',r'SGPRs Spill: (\d+)'),('o',r'Occupancy \[waves/SIMD\]: (\d+)'),('l',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,line)
        if m: cur[key]=m.group(1)
seen=set()
for r in rows:
    if r['name'] in seen: continue
    seen.add(r['name'])
    try:
        dem=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    except Exception:
        dem=r['name']
    dem=re.sub(r'\(.*','',dem).replace('(anonymous namespace)::','').replace('|','/')
    print('| %s | \`%s\` | %s | %s | %s | %s | %s | %s | %s |'%(src,dem[:90],r.get('v'),r.get('a'),r.get('s'),r.get('vs'),r.get('ss'),r.get('o'),r.get('l')))
"
done
