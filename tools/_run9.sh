timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for cfg in "t63 16" "t63 8" "t30 8"; do echo "$cfg: $(timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1 | cut -c1-120)"; done
