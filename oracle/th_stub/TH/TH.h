typedef struct THFloatTensor THFloatTensor;
float* THFloatTensor_data(THFloatTensor*);
int THFloatTensor_size(THFloatTensor*, int);
