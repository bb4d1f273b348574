#!/bin/bash
# round 4 visit 6: counters of this library's tile GEMM next to the vendor kernel on the same shapes (separate --pmc passes)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/vpmc; cd /tmp && export TMPDIR=/tmp
export VENDOR_SHAPES="46720,2048,14336;46720,2048,2048;8192,8192,8192;93312,1152,3456"
rm -rf /tmp/vp; mkdir -p /tmp/vp
timeout 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/vp/pass1 -o v -- python $R/tools/vendor_gemm_compare.py > /tmp/vp/run1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/vp/pass2 -o v -- python $R/tools/vendor_gemm_compare.py > /tmp/vp/run2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/vp/pass3 -o v -- python $R/tools/vendor_gemm_compare.py > /tmp/vp/run3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/vp/pass4 -o v -- python $R/tools/vendor_gemm_compare.py > /tmp/vp/run4.log 2>&1
cd $R
tail -4 /tmp/vp/run1.log
python tools/pmc_md_vs_vendor.py /tmp/vp "$VENDOR_SHAPES" 2>&1 | tee gpurun_out/r04_v06_md_vs_vendor_pmc.txt
