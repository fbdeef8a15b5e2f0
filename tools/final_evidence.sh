# The round's evidence on the final sources, in two halves that each fit one GPU call of twenty minutes:
#   a: the GPU suite, the counter passes and the bench line of config 4, the workspace a default batch takes
#   b: the counter passes and the bench line of config 5, the bench lines of the other workloads
# usage (on a GPU box): bash tools/final_evidence.sh [a|b]   (no argument: both)
set -e
case "${1:-ab}" in
  a) bash tools/final_evidence_a.sh ;;
  b) bash tools/final_evidence_b.sh ;;
  *) bash tools/final_evidence_a.sh && bash tools/final_evidence_b.sh ;;
esac
