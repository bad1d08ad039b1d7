cd /root/repo
for v in base prio base prio; do
  cp snerf_amd/lib/_$v.so snerf_amd/lib/libsnerf_hip.so
  echo "== $v"; timeout 300 python tools/gemm_reference_point.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); [print(k, {a:b for a,b in v.items() if 'ours' in a and 'TFLOPs' in a or a=='nt_vendor_TFLOPs'}) for k,v in d.items()]"
done
