for a in nosm nomfma nok_nov nold nosm_nok_nov_nold nosm_nomfma_nok_nov_nold; do
echo "== $a"; MLA_ATTN_LDS_EXTRA=16384 MLA_HIP_LIB=mla_amd/csrc/build_tr/libabl_$a.so MLA_ATTN_FWD=3 timeout 300 python tools/exp_attn_trace.py 2048 8 | sed -n 2,5p
done
