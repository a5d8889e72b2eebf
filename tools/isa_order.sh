#!/bin/bash
# order of loads / stores / waits / barriers / branches of one kernel (MFMAs counted): tools/isa_order.sh <isa.s> <mangled-name-regex>
awk -v pat="$2" '$0 ~ "^"pat".*:" {f=1} /^\.Lfunc_end/{if(f){exit}} f' $1 > /tmp/_k.s
grep -n "s_waitcnt vmcnt\|s_barrier\|s_cbranch\|s_endpgm\|s_branch\|global_load\|global_store\|v_mfma\|scratch_" /tmp/_k.s | awk '{print $2,$3}' | sed 's/v\[[0-9:]*\],//' | uniq -c | awk '{ if ($2 ~ /^v_mfma/) m+=$1; else { if (m) print "   mfma x", m; m=0; print } } END { if (m) print "   mfma x", m }' | grep -v "mfma x [1-3]$\|vmcnt([1-9][0-9]*)"
