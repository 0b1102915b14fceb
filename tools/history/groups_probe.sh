cd $GRAFT_REPO_ROOT
for g in 1 2 3 4; do echo "== CSADP_ROUND_GROUPS=$g"; CSADP_ROUND_GROUPS=$g python tools/msa_probe.py 2>/dev/null | grep -E "call [12]"; done
