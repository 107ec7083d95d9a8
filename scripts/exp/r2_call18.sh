timeout 300 python -m pytest tests/test_distributed.py -m gpu -q -p no:cacheprovider -k "p2p" 2>&1 | tail -3
bash scripts/profile_round.sh round2 2>&1 | tail -30
