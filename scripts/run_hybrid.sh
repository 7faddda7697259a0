#!/bin/bash
# Sweep driver with the reference's positional interface (scripts/run_hybrid.sh:3-19 in the reference):
#   bash scripts/run_hybrid.sh <dev|test> <general|legal> [--tune_linear_fusion_weight|--analyze_score_distributions]
# 11 retriever combinations x {nsf x normalisers, bcf, rrf}, one hybrid.py process each.
# Documented deviations: the third argument may be empty (the reference's README says optional, its
# script rejects it); the normaliser list is reset for every combination (the reference clobbers it after the first).
# DRY_RUN=1 prints the commands instead of running them.  Extra flags for the python driver: HYBRID_EXTRA="--synthetic 3000,16".

LLEQA_SPLIT=$1
if [ "$LLEQA_SPLIT" != "test" ] && [ "$LLEQA_SPLIT" != "dev" ]; then
    echo "ERROR: First argument corresponds to the LLeQA data split, and must be either 'test' or 'dev'."
    exit 1
fi
TRAINING_DOMAIN=$2
if [ "$TRAINING_DOMAIN" != "general" ] && [ "$TRAINING_DOMAIN" != "legal" ]; then
    echo "ERROR: Second argument corresponds to the training domain of the neural retrievers, and must be either 'general' or 'legal'."
    exit 1
fi
EXPERIMENT_NAME=$3
if [ -n "$EXPERIMENT_NAME" ] && [ "$EXPERIMENT_NAME" != "--tune_linear_fusion_weight" ] && [ "$EXPERIMENT_NAME" != "--analyze_score_distributions" ]; then
    echo "ERROR: Third argument corresponds to the experiment name, and must be either empty or one of '--tune_linear_fusion_weight' '--analyze_score_distributions'."
    exit 1
fi

HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
COMBOS=(
    "--run_bm25 --run_splade" "--run_bm25 --run_dpr" "--run_bm25 --run_colbert"
    "--run_splade --run_dpr" "--run_splade --run_colbert" "--run_dpr --run_colbert"
    "--run_bm25 --run_splade --run_dpr" "--run_bm25 --run_splade --run_colbert" "--run_bm25 --run_dpr --run_colbert"
    "--run_splade --run_dpr --run_colbert" "--run_bm25 --run_splade --run_dpr --run_colbert"
)
for R in "${COMBOS[@]}"; do
    for F in nsf bcf rrf; do
        if [ "$F" == "nsf" ]; then NORMS=("min-max" "z-score" "percentile-rank"); else NORMS=("none"); fi
        for N in "${NORMS[@]}"; do
            CMD="python $HERE/src/retrievers/hybrid.py --data_split $LLEQA_SPLIT --models_domain $TRAINING_DOMAIN $R --fusion $F --normalization $N $EXPERIMENT_NAME --output_dir output/testing $HYBRID_EXTRA"
            if [ -n "$DRY_RUN" ]; then echo "$CMD"; else $CMD || exit $?; fi
        done
    done
done
