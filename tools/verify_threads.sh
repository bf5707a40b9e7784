#!/bin/bash
# verifySendproof through the drop-in libraries from C threads (tests/dropin_driver.c verifybench: no interpreter lock between the callers): 1, 2, 4, 8, 16 threads
root=$(cd $(dirname $0)/.. && pwd); lib=$root/blockmaze_amd/lib; tmp=$(mktemp -d); exe=$tmp/drv
gcc -O1 -o $exe $root/tests/dropin_driver.c -L$lib -lzk_mint -lzk_send -lzk_deposit -lzk_redeem -lff -lsnark -lpthread -Wl,-rpath,$lib -Wl,-rpath-link,$root/blockmaze_amd || exit 1
export ZK_PRFKEY_DIR=$tmp; $root/blockmaze_amd/bin/send_key $tmp > /dev/null 2>&1 && test -s $tmp/sendpk.txt || python3 -c "
import sys; sys.path.insert(0, '$root'); from blockmaze_amd import engine as e; e.keygen('send', '$tmp/sendpk.txt', '$tmp/sendvk.txt', seed=5)" || exit 1
for t in ${@:-1 2 4 8 16}; do $exe verifybench $t 400 > /dev/null; done
rm -rf $tmp
