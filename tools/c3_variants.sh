for L in "$@"; do echo "== $L"; LERC_AMD_LIBRARY=$PWD/$L timeout 200 python tools/time_configs.py c3 2>&1 | grep -E "C3|fast_"; done
