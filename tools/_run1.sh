cd $GRAFT_REPO_ROOT
bash tools/pmc_step.sh
bash tools/pmc_one.sh "0 fwd" "0 dgrad" "0 wgrad" "1 fwd" "1 dgrad" "1 wgrad" "3 fwd" "3 dgrad" "3 wgrad" "4 fwd" "4 wgrad" "5 fwd" "5 wgrad" "7 fwd"
ls gpurun_out | grep -c pc_
