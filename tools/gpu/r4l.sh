for i in 1 2 3; do bash tools/run_variants.sh convx3h convx3h_64 convx3h_256 convx3h_512 convs2x3h 2>&1 | grep conv3x3; done
