cd $GRAFT_REPO_ROOT
for T in "" "6=0" "1=0"; do MI355_TUNE="$T" timeout 200 python scripts/diag_train_equal.py 2>&1 | grep -v amdgpu | tail -4; done
