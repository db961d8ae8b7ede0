for r in 1 2 3; do
for cfg in "t63 16" "t63 8"; do
  echo "derive $cfg: $(timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1 | cut -c1-110)"
done
done
