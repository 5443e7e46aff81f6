timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "trie_rounds" 2>&1 | tail -5
