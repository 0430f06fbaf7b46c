for v in 2 4; do python tests/_mb2.py igemm_dbg 0 $v down2,down3,down4,up3,up4 256 768; done > gpurun_out/r5_dbgw.log 2>&1
