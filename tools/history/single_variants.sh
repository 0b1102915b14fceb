cd $GRAFT_REPO_ROOT
echo "== base"; python tools/single_probe.py 16384 200000 2>/dev/null
cp csa_amd/libcsadp.so /tmp/base.so
for v in "$@"; do cp build/libcsadp_$v.so csa_amd/libcsadp.so; echo "== $v"; python tools/single_probe.py 16384 200000 2>/dev/null; done
cp /tmp/base.so csa_amd/libcsadp.so
