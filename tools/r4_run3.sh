cd $GRAFT_REPO_ROOT
for b in 8 16 32; do for n in 2 4; do echo "== batch $b, $n parts"; python tools/streamsplit.py --batch $b --parts $n --graph --iters 50 2>&1 | grep "ms per step"; done; done
