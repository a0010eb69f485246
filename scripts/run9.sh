export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
bash scripts/ab2.sh room800 "$@" | grep -v passed
RT355_LIB=$PWD/build/librt355_prof.so timeout -k 10 60 python scripts/wp2.py 2>&1 | grep RT_PROFILE | sed -n '1p;9p'
