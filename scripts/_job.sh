O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 300 python scripts/_diag_pair.py 64 1 > $O/diag_g64.log 2>&1; echo "rc=$?" >> $O/diag_g64.log
grep -q "rc=0" $O/diag_g64.log && (timeout -k 10 300 python scripts/_diag_pair.py 512 1 > $O/diag_g512.log 2>&1; echo "rc=$?" >> $O/diag_g512.log)
tail -3 $O/diag_g64.log $O/diag_g512.log
