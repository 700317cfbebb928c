cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "6 518" "21 420" "12 420"; do set -- $cfg
 for S in 1 0; do B=$1 RES=$2 SPLIT=$S python $R/tools/vit_batch_prof.py 2>&1 | grep "^B="; done
 for S in 1 0; do B=$1 RES=$2 SPLIT=$S python $R/tools/vit_batch_prof.py 2>&1 | grep "^B="; done
done
