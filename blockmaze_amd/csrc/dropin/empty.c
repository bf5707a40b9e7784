/* libff.so / libsnark.so placeholders: the cgo link line names them, nothing of theirs is referenced any more */
int zkgpu_dropin_placeholder_unused;
