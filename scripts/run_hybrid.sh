#!/bin/bash
# Sweep driver behind the reference's positional interface (its scripts/run_hybrid.sh:3-19):
#
#   bash scripts/run_hybrid.sh <dev|test> <general|legal> [--tune_linear_fusion_weight|--analyze_score_distributions]
#
# Every retriever subset of size >= 2 (11 of them, pairs first, in the reference's order) is fused three ways --
# nsf under each normaliser, then bcf and rrf without one -- with one hybrid.py process per run.
# Deviations, on purpose: the third argument may be left out (the reference's README calls it optional,
# its script refuses that), and nsf sweeps all its normalisers for EVERY subset (the reference loses the list after
# the first bcf run).
# Environment: DRY_RUN=1 prints the command lines instead of running them; HYBRID_EXTRA="--synthetic 3000,16" is appended
# to each of them.

usage_error() {
    echo "ERROR: $1"
    exit 1
}

split=$1
domain=$2
experiment=$3

case "$split" in
    dev|test) ;;
    *) usage_error "argument 1 is the LLeQA split: 'dev' or 'test' (got '${split}')." ;;
esac
case "$domain" in
    general|legal) ;;
    *) usage_error "argument 2 is the domain the neural retrievers were trained on: 'general' or 'legal' (got '${domain}')." ;;
esac
case "$experiment" in
    ""|--tune_linear_fusion_weight|--analyze_score_distributions) ;;
    *) usage_error "argument 3, if given, selects the experiment: '--tune_linear_fusion_weight' or '--analyze_score_distributions' (got '${experiment}')." ;;
esac

root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
systems=(bm25 splade dpr colbert)

# subsets of the four systems by size (2, 3, 4), members in list order: the enumeration the reference spells out by hand
subsets=()
for size in 2 3 4; do
    for mask in 3 5 9 6 10 12 7 11 13 14 15; do
        members=""; count=0
        for i in 0 1 2 3; do
            if (( (mask >> i) & 1 )); then members+=" --run_${systems[$i]}"; count=$((count + 1)); fi
        done
        if (( count == size )); then subsets+=("${members# }"); fi
    done
done

launch() {
    local cmd="python $root/src/retrievers/hybrid.py --data_split $split --models_domain $domain $1 --fusion $2 --normalization $3 $experiment --output_dir output/testing $HYBRID_EXTRA"
    if [ -n "$DRY_RUN" ]; then
        echo "$cmd"
    else
        $cmd || exit $?
    fi
}

for flags in "${subsets[@]}"; do
    for norm in min-max z-score percentile-rank; do
        launch "$flags" nsf "$norm"
    done
    launch "$flags" bcf none
    launch "$flags" rrf none
done
