for cfg in "MAP_SEED_LEN=19 MAP_ERR=0" "MAP_SEED_LEN=0 MAP_ERR=0" "MAP_SEED_LEN=19 MAP_ERR=0.02" "MAP_SEED_LEN=0 MAP_ERR=0.02"; do
  echo "== $cfg  (MAP_SEED_LEN=19: scan mode, the default; 0: the round-2 contract, seeds at the read ends only; MAP_ERR: substitution rate of the synthetic reads)"
  env $cfg timeout -s KILL 200 python tools/mapper_probe.py 2>&1 | grep -E "^index|^map"
done
