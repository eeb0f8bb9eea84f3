for w in 2 4 8 16; do echo "WPB=$w"; GR_DEV_CHROMA_WPB=$w timeout -k 10 200 python scripts/dev_systematic.py 2>&1 | grep Grids_13 | sed 's/| random.*//' ; done
