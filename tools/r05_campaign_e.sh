set -u
for c in "$@"; do
  bash tools/profile_round.sh r05_e_$c $c > gpurun_out/r05_e_${c}_log.txt 2>&1; echo "$c campaign exit $?"; tail -2 gpurun_out/r05_e_${c}_log.txt | cut -c1-200
done
