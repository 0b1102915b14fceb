cd $GRAFT_REPO_ROOT
for env in "X=1" "CSADP_BITS_CHUNK=8" "CSADP_BITS_WORDS=2"; do
  echo "== $env"; env $env python tools/single_probe.py 16384 200000 2>/dev/null
done
cp csa_amd/libcsadp.so /tmp/base.so
for v in lonepf1 lonepf2; do cp build/libcsadp_$v.so csa_amd/libcsadp.so; echo "== $v"; python tools/single_probe.py 16384 200000 2>/dev/null; done
cp /tmp/base.so csa_amd/libcsadp.so
