cd $GRAFT_REPO_ROOT
timeout -k 10 1000 bash tools/refresh_profiles.sh 2>&1 | tail -20
