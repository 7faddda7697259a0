#!/bin/bash
# BM25 driver behind the reference's positional interface (its scripts/run_bm25.sh):
#
#   bash scripts/run_bm25.sh <tuning|testing> <lleqa|mmarco>
#
# tuning : the 17 x 11 grid over (k1, b) on the validation questions -> output/tuning/bm25_tuning_results.csv (+ heat map)
# testing: one evaluation run at the dataset's preset (lleqa: k1 = 2.5, b = 0.2; mmarco: k1 = 0.9, b = 0.4) -> output/testing/
# The reference always passes --do_preprocessing (spaCy fr_core_news_md).  That model is third-party and may be absent: this script
# passes the flag when spaCy can load the model and says so on stderr when it cannot -- the text is then taken as already pre-processed.
# Environment: DRY_RUN=1 prints the command line instead of running it; BM25_EXTRA="--synthetic 3000,16" (or "--data_dir DIR") is appended.

to_perform=$1
case "$to_perform" in
    tuning|testing) ;;
    *) echo "ERROR: argument 1 is the action: 'tuning' or 'testing' (got '${to_perform}')."; exit 1 ;;
esac
dataset=$2
case "$dataset" in
    lleqa) name=lleqa ;;
    mmarco) name=mmarco-fr ;;      # (the reference hands 'mmarco' to a parser whose choices are mmarco-<lang>: the French collection is the repo's subject)
    *) echo "ERROR: argument 2 is the dataset: 'lleqa' or 'mmarco' (got '${dataset}')."; exit 1 ;;
esac

root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
prep="--do_preprocessing"
if [ -z "$DRY_RUN" ] && ! python -c "import spacy; spacy.load('fr_core_news_md')" >/dev/null 2>&1; then
    echo "run_bm25.sh: spaCy fr_core_news_md is not installed -- running WITHOUT --do_preprocessing (the text is taken as pre-processed)" >&2
    prep=""
fi

if [ "$to_perform" == "tuning" ]; then
    cmd="python $root/src/retrievers/bm25.py --dataset $name $prep --do_hyperparameter_tuning --output_dir output/tuning $BM25_EXTRA"
else
    if [ "$dataset" == "lleqa" ]; then k1=2.5; b=0.2; else k1=0.9; b=0.4; fi
    cmd="python $root/src/retrievers/bm25.py --dataset $name --do_evaluation $prep --k1 $k1 --b $b --output_dir output/testing $BM25_EXTRA"
fi
if [ -n "$DRY_RUN" ]; then
    echo "$cmd"
else
    $cmd || exit $?
fi
