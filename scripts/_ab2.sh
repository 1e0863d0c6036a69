for sp in 1 2 4 8; do QSPEC_ATTN_SPLITS=$sp python scripts/profile_cycle.py --steps 20 --model tinyllama-1.1b --batch 1 --k 3 2>/dev/null | cut -c1-20 | sed "s/^/splits=$sp /"; done
for sp in 1 2 4 8; do QSPEC_ATTN_SPLITS=$sp python scripts/profile_cycle.py --steps 20 --model tinyllama-1.1b --batch 1 --k 3 --ctx 1900 2>/dev/null | cut -c1-20 | sed "s/^/ctx1900 splits=$sp /"; done
