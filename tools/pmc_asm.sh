cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcasm; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/p1 -- python3 tools/probe_asm.py c4_10m > $out/p1.log 2>&1
python tools/pmc_table.py $(ls $out/p1/*/*counter_collection.csv | head -1) | grep -i assemble
rm -rf $out/p1
