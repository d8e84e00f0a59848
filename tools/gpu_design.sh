cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_design_driver.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import csv
tg={r["name"]:r["structure"] for r in csv.DictReader(open("tests/golden/eterna_v1_targets.csv"))}
open("/tmp/ete69.txt","w").write(">name\nEte_69\n>seq_restr\n%s\n>sec_struct\n%s\n"%("N"*200, tg["eteV1_69.txt"]))
PY
echo "== native host loop"; timeout 600 python -m desirna_amd.design -f /tmp/ete69.txt -R 64 -e 100 -s 5 -seed 1 2>&1 | tail -4
echo "== python host loop"; timeout 600 python -m desirna_amd.design -f /tmp/ete69.txt -R 64 -e 100 -s 5 -seed 1 --python-host 2>&1 | tail -4
