# end-to-end lines of profiles/: tools/e2e_synthetic.py (simulate -> scripts/train.py -> novel-view PSNR), the same with the dynamic
# batch size rounded to 1 024 events (captured steps), and the kernel stats + step sequence of the training run's steady state
#   gpurun --timeout 1500 -- 'bash tools/regen_e2e.sh r06'
RND=${1:-r06}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
python tools/e2e_synthetic.py --out $O/e2e > $O/e2e.log 2>&1
cp $O/e2e/e2e_result.json profiles/${RND}_e2e_result.json; cp $O/e2e/train.yaml profiles/${RND}_e2e_train.yaml; cp $O/e2e/e2e_novel_views.png profiles/${RND}_e2e_novel_views.png
python tools/e2e_synthetic.py --out $O/e2e_q --batch-size-quantum 1024 > $O/e2e_q.log 2>&1
cp $O/e2e_q/e2e_result.json profiles/${RND}_e2e_result_batch_quantum_1024.json
rm -rf $O/e2e/dataset $O/e2e/init $O/e2e/run $O/e2e_q/dataset $O/e2e_q/init $O/e2e_q/run
bash tools/e2e_kstats.sh profiles/${RND}_e2e_kernel_stats.csv > $O/e2e_kstats.log 2>&1
mv profiles/${RND}_e2e_kernel_stats_gaps.txt profiles/${RND}_e2e_step_sequence.txt
cp profiles/${RND}_e2e_* $O/          # (gpurun brings gpurun_out/ back, not profiles/)
tail -5 $O/e2e.log; grep "train_log_tail" -A 7 profiles/${RND}_e2e_result_batch_quantum_1024.json | cut -c1-200; grep -n "scan_guard" profiles/${RND}_e2e_kernel_stats.csv
