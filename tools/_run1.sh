for i in 1 2; do timeout -k 10 300 python3 tools/_graph_var.py 2>&1 | tail -15; done
