cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_design_driver.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
cat > /tmp/ete69.txt <<'EOT'
>name
Ete_69
>seq_restr
NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN
>sec_struct
.((.((.....(((((...((.(((.........................))).))...))).))...)).)).(((..(((..((.......((....)).......)))))......))).(((((.......(((..((..(((..((.(((...............)))..))..)))..)).)))..))..))).
EOT
timeout 600 python -m desirna_amd.design -f /tmp/ete69.txt -R 64 -e 100 -s 3 -seed 1 2>&1 | tail -5
