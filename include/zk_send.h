/* Drop-in for libsnark-vnt/src/send/sendcgo.hpp. */
#ifndef ZK_SEND_H
#define ZK_SEND_H
#include "zk_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* replaces libsnark-vnt/src/send/sendcgo.cpp:256-272 (also exported by libzk_deposit.so) */
char *genCMTS(uint64_t value_s, char *pk_recv_string, char *r_s_string, char *sn_old_string);
/* replaces libsnark-vnt/src/send/sendcgo.cpp:288-299 */
char *computeCRH(char *pk_string, char *r_string);
/* replaces libsnark-vnt/src/send/sendcgo.cpp:301-364; key file /usr/local/prfKey/sendpk.txt.  Called by GenSendProof (go-ethereum/zktx/zktx.go:406-430). */
char *genSendproof(uint64_t value_A, char *r_s_string, char *sn_string, char *r_string, char *cmt_s_string, char *cmtA_string, uint64_t value_s,
                   char *pk_recv_string, uint64_t value_A_new, char *sn_A_new, char *r_A_new, char *cmt_A_new, char *sk_string, char *pk_sender_string);
/* replaces libsnark-vnt/src/send/sendcgo.cpp:366-463; key file /usr/local/prfKey/sendvk.txt */
bool verifySendproof(char *data, char *cmtA_old_string, char *sn_old_string, char *cmtS_string, char *cmtA_new_string);
#ifdef __cplusplus
}
#endif
#endif
