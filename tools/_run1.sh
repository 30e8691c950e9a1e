for i in 1 2 3; do timeout -k 10 380 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
