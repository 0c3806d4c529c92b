# compact schedule map of a kernel's ISA: M mfma, d ds_read, w ds_write, G lds-dma, L load, S store, a accvgpr, T transcendental, v other VALU
awk -v A=$2 -v B=$3 'NR>=A && NR<=B' $1 | grep -v "^[[:space:]]*;" | awk '
/v_mfma/ {printf "M"; next}
/ds_read/ {printf "d"; next}
/ds_write/ {printf "w"; next}
/global_load_lds/ {printf "G"; next}
/global_load/ {printf "L"; next}
/global_store/ {printf "S"; next}
/s_waitcnt vmcnt/ {printf "[%s]", $2; next}
/s_waitcnt lgkmcnt/ {printf "|"; next}
/s_waitcnt/ {printf "[%s %s]", $2,$3; next}
/s_barrier/ {printf "\nBARRIER\n"; next}
/v_accvgpr/ {printf "a"; next}
/v_exp|v_rcp/ {printf "T"; next}
/^[[:space:]]*v_/ {printf "v"; next}
/^[[:space:]]*s_nop/ {printf "n"; next}
/^[[:space:]]*s_/ {printf "."; next}
/^.LBB/ {printf "\n%s\n", $1; next}
' | fold -w 200
