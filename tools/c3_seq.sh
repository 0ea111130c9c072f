# config 3 (bench.py --report): loss phases + the launch sequence of one step -> gpurun_out/c3_phases.txt, gpurun_out/kseq_c3.txt
cd $GRAFT_REPO_ROOT
python tools/loss_phases.py 10 > gpurun_out/c3_phases.txt 2>&1
bash tools/kseq.sh c3 --report
