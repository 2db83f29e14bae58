for f in 0 1; do ELLP_NO_UNIT_COLUMNS=$([ $f = 1 ] && echo 1) ; if [ $f = 1 ]; then export ELLP_NO_UNIT_COLUMNS=1; else unset ELLP_NO_UNIT_COLUMNS; fi
python bench.py --solver dual --config4 0 --config5 0 --full-solve 0 --cpu-pivots 0 --same-alg-pivots 0 --long-window 0 --steps 2000 --warmup 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no_unit_columns=$f', d['value'], d['ms_per_step'], d['kernels_us'])"
done
