R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sage_eager_x -- python3 $R/tools/sage_profile.py eager 60 > $R/gpurun_out/sage_eager_x.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/sage_eager_x 60 | head -22
python -m pytest tests/test_epilogue_gpu.py tests/test_sage_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
