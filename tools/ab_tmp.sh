for nw in 4 8; do
  echo "== FIL_CIN_QS_FWD_WAVES=$nw"
  FIL_CIN_QS_FWD_WAVES=$nw timeout 300 python tools/qsplit_check.py 2>&1 | grep -E "mode 2"
done
