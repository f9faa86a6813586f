run4() {
  for k in 1 2 3; do (python tools/fresh_context_stress.py 60 > /tmp/s$k.txt 2>&1 &); done
  python tools/fresh_context_stress.py 60 2>&1 | tail -6
  sleep 8
  for k in 1 2 3; do tail -n 1 /tmp/s$k.txt; done
}
echo "--- alone"; python tools/fresh_context_stress.py 60 2>&1 | tail -4
echo "--- four processes"; run4
