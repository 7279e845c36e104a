for k in 0 1 2 3 4 6 8; do echo "log_k=$k"; PM_TUNE="log_k=$k" PYTHONPATH=. timeout 120 python tools/exp_conv.py 4096 2>&1 | grep float32; done
